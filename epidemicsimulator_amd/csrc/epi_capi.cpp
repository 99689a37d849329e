// Host engine and C ABI (include/epi.h).  Builds the per-sweep tables with the reference's
// operation order (compiled with -ffp-contract=off), owns the device buffers and enqueues the
// kernels of epi_kernels.hip.  There is no CPU compute path: every entry point that computes
// needs a HIP device and fails with EPI_ERR_DEVICE without one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/epi.h"
#include "epi_host.hpp"
#include "epi_kernels.hpp"
#include "java_compat.hpp"

#define EPI_API extern "C" __attribute__((visibility("default")))

namespace {

using namespace epi;

thread_local std::string g_last_error;
thread_local int32_t g_last_code = 0;

}  // namespace

namespace epi {
int32_t fail(int32_t code, const char* fmt, ...) {
    g_last_code = code;
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
}  // namespace epi

namespace {

const epi_profile kDefaultProfile = {1.0, 1.0, 0.05, 100.0, 30000.0, 0.7, 0.15, 0.8, 0.7};  // Geo.scala:4-21

Hyper to_hyper(const epi_hyper& h) {
    Hyper o;
    o.epsilon_start = h.epsilon_start; o.epsilon_end = h.epsilon_end;
    o.epsilon_decay_steps = h.epsilon_decay_steps; o.hidden = h.hidden;
    o.clip_q = h.clip_q; o.reward_clip = h.reward_clip; o.batch_size = h.batch_size; o.learn_every = h.learn_every;
    return o;
}

bool is_count(double x) { return std::isfinite(x) && x >= 0.0 && x <= 2147483647.0 && std::floor(x) == x; }

// ToyConn.scaleRow on one row (ToyConn.scala:58-70).
void scale_row_inplace(double* row, int n, int self, double factor) {
    double s = 0.0;
    for (int j = 0; j < n; ++j)
        if (j != self) { row[j] *= factor; s += row[j]; }
    const double cap = std::fmin(0.2, s);
    if (s > cap && s > 0) {
        const double sc = cap / s;
        for (int j = 0; j < n; ++j)
            if (j != self) row[j] *= sc;
    }
    row[self] = 0.0;
}

// Dense row-major copy of the connection matrix from either input form.
int32_t densify(int n, const double* conn, const int32_t* rowptr, const int32_t* col, const double* val,
                std::vector<double>& W) {
    W.assign((size_t)n * n, 0.0);
    if (conn) {
        std::memcpy(W.data(), conn, sizeof(double) * (size_t)n * n);
        return EPI_OK;
    }
    if (!rowptr || !col || !val) return fail(EPI_ERR_BAD_ARGUMENT, "neither a dense matrix nor a CSR triple was given");
    for (int i = 0; i < n; ++i) {
        int prev = -1;
        for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            const int j = col[k];
            if (j < 0 || j >= n || j <= prev)
                return fail(EPI_ERR_BAD_ARGUMENT, "CSR row %d: column indices must be ascending and in [0,n)", i);
            prev = j;
            W[(size_t)i * n + j] = val[k];
        }
    }
    return EPI_OK;
}

// Device-resident tables shared by the sweep and the world API.
// Elements a long row's compact copy occupies: whole groups of 64 (1, 2 or 4 groups for the register
// routines up to 256 weights, as many as needed for the streamed rows above that).
static size_t long_row_slots(int deg) {
    if (deg <= kMaxSparseDegree) return (size_t)64 * dense_slots_for(deg);
    return ((size_t)deg + 63) / 64 * 64;
}

struct GraphTables {
    int n = 0, npad = 0;
    DevBuf q, sumw, pop, qtop;
    DevBuf d_W;             // dense graphs: the matrix itself, kept while the handle's tables are being built
    std::vector<double> W;  // dense host copy (checked for finite, non-negative weights before the upload)
    // sparse mode (n > kMaxDenseN): CSR arrays on the device, raw weights on the host
    bool sparse = false;
    int max_deg = 0, lane_deg = 0;
    DevBuf d_rowptr, d_col, d_qval, d_ell, d_ell_rows, d_ell_pos, d_long_rows, d_long_off, d_long_q, d_long_col;
    std::vector<int32_t> h_long_rows;     // rows longer than kLaneSlots
    std::vector<int32_t> h_slots, h_off;  // ELL block tables
    std::vector<int32_t> h_pos;           // row -> position in the lane layout (block * 64 + lane), -1 for long rows
    int n_blocks = 0;
    std::vector<int32_t> h_rowptr, h_col;
    std::vector<double> h_val;
    // A sweep numbers the patches of a sparse graph in lane-layout order inside the engine (renumbered): internal id
    // p < n_ell is the row at position p of the length-sorted blocks, the long rows follow.  A block of 64 rows is then
    // 64 consecutive patches: the mobility phase reads the exports and the local phase writes them with coalesced
    // accesses, and no second, position-ordered copy of the exports exists.  id_of[c] = internal id of the caller's
    // patch c, orig_of = its inverse; both empty when the numbering is the caller's.
    bool renumbered = false;
    int n_ell = 0;
    std::vector<int32_t> id_of, orig_of;
    DevBuf d_id_of;
    int inner(int c) const { return renumbered ? id_of[(size_t)c] : c; }
    int outer(int p) const { return renumbered && p >= 0 && p < n ? orig_of[(size_t)p] : p; }

    // Normalises one stored row in column order: the reference sums the full row left to right, and
    // adding the zeros of the entries that are not stored changes no partial sum.
    static int32_t normalise_sparse_row(const double* val, const int32_t* col, int deg, int self, double* q_out,
                                        double* sumw_out) {
        double sum = 0.0;
        for (int k = 0; k < deg; ++k) {
            const double w = (col[k] == self) ? 0.0 : val[k];
            if (!std::isfinite(w) || w < 0.0)
                return fail(EPI_ERR_BAD_ARGUMENT, "connection weight (%d,%d) = %g: weights must be finite and >= 0", self, col[k], w);
            sum = (k == 0) ? w : sum + w;
        }
        *sumw_out = std::fmax(0.0, sum);
        for (int k = 0; k < deg; ++k) {
            const double w = (col[k] == self) ? 0.0 : val[k];
            q_out[k] = (sum > 0.0) ? w / sum : 0.0;
        }
        return EPI_OK;
    }

    // Weights of one ELL block (slot-major, whole slot groups) read back from the device: the seed row's block, of
    // which a sweep keeps one copy per travel scale.
    int32_t download_block_q(int b, std::vector<double>& out) const {
        const size_t slots = (size_t)(h_slots[(size_t)b] + kSlotGroup - 1) / kSlotGroup * kSlotGroup;
        out.assign(slots * 64, 0.0);
        if (slots == 0) return EPI_OK;
        HIP_TRY(hipMemcpy(out.data(), d_qval.as<double>() + (size_t)h_off[(size_t)b] * 64, slots * 64 * sizeof(double), hipMemcpyDeviceToHost));
        return EPI_OK;
    }

    int32_t build_sparse(const int32_t* pop_, const double* conn, const int32_t* rowptr, const int32_t* col,
                         const double* val, bool renumber) {
        sparse = true;
        npad = (n + 63) / 64 * 64;
        if (csr_lds_bytes(n) > kMaxLdsBytes)
            return fail(EPI_ERR_UNSUPPORTED, "n = %d patches: the sparse kernel keeps 8 bytes of inflow per patch in the %zu KiB of LDS", n, kMaxLdsBytes / 1024);
        // host copy of the graph in CSR form: checked here, and the seed-row variants of a sweep are cut from it
        h_rowptr.assign((size_t)n + 1, 0);
        h_col.clear();
        h_val.clear();
        if (rowptr && col && val) {
            if (rowptr[n] < rowptr[0]) return fail(EPI_ERR_BAD_ARGUMENT, "CSR row pointers must be ascending");
            for (int i = 0; i < n; ++i) {
                if (rowptr[i + 1] < rowptr[i]) return fail(EPI_ERR_BAD_ARGUMENT, "CSR row pointers must be ascending (row %d)", i);
                int prev = -1;
                for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
                    if (col[k] < 0 || col[k] >= n || col[k] <= prev)
                        return fail(EPI_ERR_BAD_ARGUMENT, "CSR row %d: column indices must be ascending and in [0,n)", i);
                    prev = col[k];
                }
                h_rowptr[(size_t)i + 1] = rowptr[i + 1] - rowptr[0];
            }
            h_col.assign(col + rowptr[0], col + rowptr[n]);
            h_val.assign(val + rowptr[0], val + rowptr[n]);
        } else if (conn) {
            for (int i = 0; i < n; ++i) {
                for (int j = 0; j < n; ++j)
                    if (conn[(size_t)i * n + j] != 0.0 && j != i) { h_col.push_back(j); h_val.push_back(conn[(size_t)i * n + j]); }
                h_rowptr[i + 1] = (int32_t)h_col.size();
            }
        } else {
            return fail(EPI_ERR_BAD_ARGUMENT, "neither a dense matrix nor a CSR triple was given");
        }
        max_deg = 0;
        for (int i = 0; i < n; ++i) {
            const int rs = h_rowptr[i], re = h_rowptr[i + 1];
            max_deg = std::max(max_deg, re - rs);
            for (int k = rs; k < re; ++k) {
                const double w = h_val[(size_t)k];
                if (h_col[(size_t)k] != i && (!std::isfinite(w) || w < 0.0))
                    return fail(EPI_ERR_BAD_ARGUMENT, "connection weight (%d,%d) = %g: weights must be finite and >= 0", i, h_col[(size_t)k], w);
            }
        }
        for (int c = 0; c < n; ++c)
            if (pop_[c] < 0) return fail(EPI_ERR_BAD_ARGUMENT, "pop[%d] = %d is negative", c, pop_[c]);
        const size_t nnz = h_col.size();
        // Row order on the host (lengths only, a stable counting sort: O(n)): the rows a lane can hold sorted by length,
        // longest first, rows of one length in index order, cut into blocks of 64 -- a block then stores (and the kernel
        // reads and sorts) about as many slots as its rows really have, and the work queue hands out the expensive
        // blocks first; the longer rows follow in index order.
        std::vector<int32_t> order((size_t)n);
        {
            std::vector<int32_t> start((size_t)kLaneSlots + 2, 0);
            auto bucket = [&](int i) { const int dg = h_rowptr[i + 1] - h_rowptr[i]; return dg > kLaneSlots ? kLaneSlots + 1 : dg; };
            for (int i = 0; i < n; ++i) start[(size_t)bucket(i)] += 1;
            int pos = 0;
            std::vector<int32_t> cursor((size_t)kLaneSlots + 2, 0);
            for (int dg = kLaneSlots; dg >= 0; --dg) { cursor[(size_t)dg] = pos; pos += start[(size_t)dg]; }
            n_ell = pos;
            cursor[(size_t)kLaneSlots + 1] = pos;
            for (int i = 0; i < n; ++i) order[(size_t)cursor[(size_t)bucket(i)]++] = i;
        }
        const int n_long = n - n_ell;
        n_blocks = (n_ell + 63) / 64;
        h_slots.assign((size_t)(n_blocks > 0 ? n_blocks : 1), 0);
        h_off.assign((size_t)(n_blocks > 0 ? n_blocks : 1), 0);
        int total_slots = 0;
        for (int b2 = 0; b2 < n_blocks; ++b2) {
            const int first = order[(size_t)b2 * 64];
            const int dmax = h_rowptr[first + 1] - h_rowptr[first];   // a block's longest row is its first one
            h_slots[(size_t)b2] = dmax;
            h_off[(size_t)b2] = total_slots;
            total_slots += (dmax + kSlotGroup - 1) / kSlotGroup * kSlotGroup;   // whole slot groups, zero filled
        }
        lane_deg = n_blocks > 0 ? h_slots[0] : 0;
        h_long_rows.assign(order.begin() + n_ell, order.end());
        h_pos.assign((size_t)n, -1);
        for (int p = 0; p < n_ell; ++p) h_pos[(size_t)order[(size_t)p]] = p;
        // A sweep numbers the patches in this order inside the engine (see `renumbered`)
        renumbered = renumber;
        id_of.assign((size_t)n, 0);
        for (int p = 0; p < n; ++p) id_of[(size_t)order[(size_t)p]] = p;
        orig_of.clear();
        if (renumbered) orig_of = order;
        std::vector<int32_t> lo((size_t)(n_long ? n_long : 1), 0);
        size_t long_elems = 0;
        for (int o = 0; o < n_long; ++o) {
            const int i = h_long_rows[(size_t)o];
            lo[(size_t)o] = (int32_t)long_elems;
            long_elems += long_row_slots(h_rowptr[i + 1] - h_rowptr[i]);
        }
        const size_t long_alloc = long_elems ? long_elems : 64;
        const size_t ell_alloc = (size_t)(total_slots > 0 ? total_slots : 1) * 64;
        // The tables the kernels read are allocated first, the staging of the build after them: where a buffer lands
        // decides a few per cent of the sweep's speed, and this order is the one the measurements were made with.
        HIP_TRY(d_long_off.upload(lo));
        HIP_TRY(d_long_rows.ensure(sizeof(int32_t) * (size_t)(n_long ? n_long : 1)));
        HIP_TRY(d_long_q.ensure(sizeof(double) * long_alloc));
        HIP_TRY(d_long_col.ensure(sizeof(int32_t) * long_alloc));
        {   // row lengths in the engine's numbering (the kernel reads them for the long rows)
            std::vector<int32_t> rp((size_t)n + 1, 0);
            for (int p = 0; p < n; ++p) {
                const int c = order[(size_t)p];
                rp[(size_t)(renumbered ? p : c) + 1] = h_rowptr[c + 1] - h_rowptr[c];
            }
            for (int p = 0; p < n; ++p) rp[(size_t)p + 1] += rp[(size_t)p];
            HIP_TRY(d_rowptr.upload(rp));
        }
        std::vector<EllBlock> hb(h_slots.size());
        for (size_t b2 = 0; b2 < hb.size(); ++b2) {
            hb[b2].slots = h_slots[b2];
            hb[b2].off = h_off[b2];
        }
        HIP_TRY(d_ell.upload(hb));
        HIP_TRY(d_id_of.upload(id_of));
        if (!renumbered) id_of.clear();
        HIP_TRY(d_col.ensure(sizeof(int32_t) * ell_alloc));
        HIP_TRY(d_qval.ensure(sizeof(double) * ell_alloc));
        HIP_TRY(sumw.ensure(sizeof(double) * (size_t)npad));
        HIP_TRY(pop.ensure(sizeof(int32_t) * (size_t)npad));
        {   // row of every lane of every block, -1 = no row (in the engine's numbering when the patches are renumbered:
            // position p then holds patch p itself)
            std::vector<int32_t> rows((size_t)(n_blocks > 0 ? n_blocks : 1) * 64, -1);
            if (renumbered) for (int p = 0; p < n_ell; ++p) rows[(size_t)p] = p;
            else std::copy(order.begin(), order.begin() + n_ell, rows.begin());
            HIP_TRY(d_ell_rows.upload(rows));
        }
        if (renumbered) {
            d_ell_pos.release();
        } else {
            std::vector<int32_t> pos_pad((size_t)npad, -1);
            std::copy(h_pos.begin(), h_pos.end(), pos_pad.begin());
            HIP_TRY(d_ell_pos.upload(pos_pad));
        }
        HIP_TRY(hipMemset(d_long_rows.p, 0, sizeof(int32_t) * (size_t)(n_long ? n_long : 1)));
        HIP_TRY(hipMemset(d_long_q.p, 0, sizeof(double) * long_alloc));
        HIP_TRY(hipMemset(d_long_col.p, 0xff, sizeof(int32_t) * long_alloc));
        HIP_TRY(hipMemset(sumw.p, 0, sizeof(double) * (size_t)npad));
        HIP_TRY(hipMemset(pop.p, 0, sizeof(int32_t) * (size_t)npad));
        // The arithmetic and the big copies happen on the device: row sums and normalised weights from the caller's CSR
        // arrays, then the slot-major blocks of 64 rows and the compact copies of the longer rows.
        DevBuf c_rp, c_col, c_val, c_pop, qcsr, sumw_rows, d_order;
        HIP_TRY(c_rp.upload(h_rowptr));
        HIP_TRY(c_col.ensure(sizeof(int32_t) * (nnz ? nnz : 1)));
        HIP_TRY(c_val.ensure(sizeof(double) * (nnz ? nnz : 1)));
        if (nnz) {
            HIP_TRY(hipMemcpy(c_col.p, h_col.data(), sizeof(int32_t) * nnz, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(c_val.p, h_val.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
        }
        HIP_TRY(c_pop.upload(std::vector<int32_t>(pop_, pop_ + n)));
        HIP_TRY(qcsr.ensure(sizeof(double) * (nnz ? nnz : 1)));
        HIP_TRY(sumw_rows.ensure(sizeof(double) * (size_t)n));
        HIP_TRY(d_order.upload(order));
        SparseBuildArgs sa{};
        sa.n = n; sa.npad = npad;
        sa.rowptr = c_rp.as<int32_t>(); sa.col = c_col.as<int32_t>(); sa.val = c_val.as<double>();
        sa.qcsr = qcsr.as<double>(); sa.sumw_rows = sumw_rows.as<double>();
        sa.order = d_order.as<int32_t>(); sa.n_ell = n_ell; sa.blocks = d_ell.as<EllBlock>();
        sa.col_map = renumbered ? d_id_of.as<int32_t>() : nullptr;
        sa.pop_in = c_pop.as<int32_t>(); sa.pop_out = pop.as<int32_t>(); sa.sumw_out = sumw.as<double>();
        sa.ell_q = d_qval.as<double>(); sa.ell_col = d_col.as<int32_t>();
        sa.n_long = n_long; sa.long_off = d_long_off.as<int32_t>();
        sa.long_q = d_long_q.as<double>(); sa.long_col = d_long_col.as<int32_t>(); sa.long_rows_out = d_long_rows.as<int32_t>();
        HIP_TRY(launch_sparse_normalise(sa, nullptr));
        HIP_TRY(launch_sparse_fill(sa, n_blocks, nullptr));
        HIP_TRY(hipDeviceSynchronize());   // the temporary buffers go out of scope
        return EPI_OK;
    }

    void fill(EpisodeArgs& ea) const {
        ea.sumw = sumw.as<double>();
        if (sparse) {
            ea.q = nullptr; ea.qtop = nullptr;
            ea.rowptr = d_rowptr.as<int32_t>(); ea.col = d_col.as<int32_t>(); ea.qval = d_qval.as<double>();
            ea.ell = d_ell.as<EllBlock>();
            ea.ell_rows = renumbered ? nullptr : d_ell_rows.as<int32_t>();
            ea.block_rows = d_ell_rows.as<int32_t>();
            ea.ell_pos = renumbered ? nullptr : d_ell_pos.as<int32_t>();
            ea.n_blocks = n_blocks;
            ea.n_ell = n_ell;
            ea.seed_block = -1;
            ea.max_deg = lane_deg;
            ea.n_long = (int)h_long_rows.size();
            ea.long_rows = d_long_rows.as<int32_t>(); ea.long_off = d_long_off.as<int32_t>(); ea.long_q = d_long_q.as<double>(); ea.long_col = d_long_col.as<int32_t>();
        } else {
            ea.q = q.as<double>(); ea.qtop = qtop.as<double>();
            ea.rowptr = nullptr;
        }
    }

    int32_t build(int n_, const int32_t* pop_, const double* conn, const int32_t* rowptr, const int32_t* col,
                  const double* val, bool renumber = false) {
        n = n_;
        if (n > kMaxDenseN) return build_sparse(pop_, conn, rowptr, col, val, renumber);
        npad = dense_slots_for(n) * 64;
        int32_t rc = densify(n, conn, rowptr, col, val, W);
        if (rc) return rc;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const double w = W[(size_t)i * n + j];
                if (j != i && (!std::isfinite(w) || w < 0.0))
                    return fail(EPI_ERR_BAD_ARGUMENT, "connection weight (%d,%d) = %g: weights must be finite and >= 0", i, j, w);
            }
        std::vector<int32_t> hp((size_t)npad, 0);
        for (int c = 0; c < n; ++c) {
            if (pop_[c] < 0) return fail(EPI_ERR_BAD_ARGUMENT, "pop[%d] = %d is negative", c, pop_[c]);
            hp[c] = pop_[c];
        }
        // the tables are built on the device from the uploaded matrix (prepare_rows_kernel): normalised rows, row
        // sums, top-weight lists
        HIP_TRY(d_W.upload(W));
        HIP_TRY(q.ensure(sizeof(double) * (size_t)n * npad));
        HIP_TRY(qtop.ensure(sizeof(double) * (size_t)n * kTopWeights));
        HIP_TRY(sumw.ensure(sizeof(double) * (size_t)npad));
        HIP_TRY(hipMemsetAsync(sumw.p, 0, sizeof(double) * (size_t)npad, nullptr));
        PrepareRowsArgs pa{};
        pa.n = n; pa.npad = npad; pa.W = d_W.as<double>(); pa.n_rows = n; pa.seed_idx = 0; pa.travel = nullptr;
        pa.q = q.as<double>(); pa.sumw = sumw.as<double>(); pa.qtop = qtop.as<double>();
        HIP_TRY(launch_prepare_rows(pa, nullptr));
        HIP_TRY(pop.upload(hp));
        return EPI_OK;
    }
};

// Per-episode work buffers.
struct EpisodeBuffers {
    int count = 0, npad = 0;
    DevBuf st, cap, tcur, arrival, peak, err, act_ep, exports;
    int32_t ensure(int count_, int npad_, bool sparse = false) {
        if (count_ <= count && npad_ == npad && (!sparse || exports.p)) return EPI_OK;
        const size_t cells = (size_t)count_ * npad_;
        if (sparse && cells * 5 * sizeof(int32_t) > 0xffffffffull)  // the sparse kernel reaches a launch's planes through 32-bit byte offsets
            return fail(EPI_ERR_UNSUPPORTED, "%d instances of %d padded patches in one launch: a sparse world takes at most %zu at a time",
                        count_, npad_, (size_t)(0xffffffffull / (5 * sizeof(int32_t) * (size_t)npad_)));
        HIP_TRY(st.ensure(cells * 5 * sizeof(int32_t)));
        HIP_TRY(cap.ensure(cells * sizeof(double)));
        HIP_TRY(tcur.ensure(cells * sizeof(int32_t)));
        HIP_TRY(arrival.ensure(cells * sizeof(int32_t)));
        HIP_TRY(peak.ensure(cells * sizeof(int32_t)));
        HIP_TRY(err.ensure((size_t)count_ * 4 * sizeof(int32_t)));
        HIP_TRY(act_ep.ensure(cells));
        if (sparse) HIP_TRY(exports.ensure(cells * 4 * sizeof(int32_t)));
        count = count_;
        npad = npad_;
        return EPI_OK;
    }
};

// Raw addresses of one set of per-episode work buffers: the handle's own allocations (EpisodeBuffers) or a
// carve-up of the fixed scratch array below.
struct WorkPtrs {
    int32_t* st = nullptr; double* cap = nullptr; int32_t* tcur = nullptr; int32_t* arrival = nullptr;
    int32_t* peak = nullptr; int32_t* err = nullptr; int8_t* act_ep = nullptr; int32_t* exports = nullptr;
};

WorkPtrs work_ptrs(const EpisodeBuffers& w) {
    WorkPtrs p;
    p.st = w.st.as<int32_t>(); p.cap = w.cap.as<double>(); p.tcur = w.tcur.as<int32_t>();
    p.arrival = w.arrival.as<int32_t>(); p.peak = w.peak.as<int32_t>(); p.err = w.err.as<int32_t>();
    p.act_ep = w.act_ep.as<int8_t>(); p.exports = w.exports.as<int32_t>();
    return p;
}

// ---- fixed work area of epi_grid_run_noalloc -----------------------------------------------------------------
// Device memory that exists from the moment the library is loaded until the process ends, one copy per GPU.
// On the host `g_noalloc_scratch` is only a handle for hipGetSymbolAddress; the device address is looked up once
// per device (scratch_base) and that address is what kernels and copies receive.
constexpr size_t kNoallocScratchBytes = (size_t)32 << 20;
constexpr int kMaxScratchDevices = 64;
static __device__ __align__(256) unsigned char g_noalloc_scratch[kNoallocScratchBytes];
static void* g_scratch_dev[kMaxScratchDevices];   // per-device address of the array, null until looked up
static std::mutex g_scratch_mu;

// Address of the calling thread's current device's copy of the scratch array.
int32_t scratch_base(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxScratchDevices)
        return fail(EPI_ERR_UNSUPPORTED, "device %d: the fixed work area is kept for devices 0..%d", device, kMaxScratchDevices - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_dev[device]) {
        void* p = nullptr;
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch)));
        if (!p) return fail(EPI_ERR_DEVICE, "hipGetSymbolAddress returned a null address for the fixed work area");
        g_scratch_dev[device] = p;
    }
    *out = static_cast<unsigned char*>(g_scratch_dev[device]);
    return EPI_OK;
}

// Bytes of work area `count` episodes of `npad` padded countries need, every section rounded up to 256 bytes.
size_t scratch_need(size_t count, size_t npad, bool sparse) {
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t cells = count * npad;
    size_t need = up(cells * 8) + up(cells * 20) + 3 * up(cells * 4) + up(count * 16) + up(cells);
    if (sparse) need += up(cells * 16);
    return need;
}

WorkPtrs carve_scratch(unsigned char* base, size_t count, size_t npad, bool sparse) {
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t cells = count * npad;
    WorkPtrs p;
    unsigned char* q = base;
    p.cap = reinterpret_cast<double*>(q); q += up(cells * 8);
    p.st = reinterpret_cast<int32_t*>(q); q += up(cells * 20);
    p.tcur = reinterpret_cast<int32_t*>(q); q += up(cells * 4);
    p.arrival = reinterpret_cast<int32_t*>(q); q += up(cells * 4);
    p.peak = reinterpret_cast<int32_t*>(q); q += up(cells * 4);
    p.err = reinterpret_cast<int32_t*>(q); q += up(count * 16);
    p.act_ep = reinterpret_cast<int8_t*>(q); q += up(cells);
    if (sparse) { p.exports = reinterpret_cast<int32_t*>(q); q += up(cells * 16); }
    return p;
}

struct MlpDevice {
    NetLayout nl{0, 0};
    double clip_q = 10.0;
    DevBuf net;   // one network in NetLayout form
    int32_t upload(int h, const double* w1, const double* bb1, const double* w2, const double* bb2, const double* w3,
                   const double* bb3, double clip) {
        if (h < 1 || h > 256) return fail(EPI_ERR_UNSUPPORTED, "hidden = %d: the device network supports 1..256 units", h);
        nl.H = h;
        nl.Hp = (h + 63) / 64 * 64;
        clip_q = clip;
        std::vector<double> img(nl.size());
        pack_net(nl, w1, bb1, w2, bb2, w3, bb3, img.data());
        HIP_TRY(net.upload(img));
        HIP_TRY(screen_stats.ensure(2 * sizeof(unsigned long long)));   // states screened / flagged since this upload
        HIP_TRY(hipMemset(screen_stats.p, 0, 2 * sizeof(unsigned long long)));
        // constants of the screen pass's error bound (MlpArgs::cert): largest absolute row sum and bias per layer
        auto norms = [](const double* w, const double* b, int rows, int cols, double& A, double& B) {
            A = B = 0.0;
            for (int r = 0; r < rows; ++r) {
                double sum = 0.0;
                for (int c = 0; c < cols; ++c) sum += std::fabs(w[(size_t)r * cols + c]);
                A = std::fmax(A, sum);
                if (b) B = std::fmax(B, std::fabs(b[r]));
            }
        };
        norms(w1, bb1, h, 7, cert[0], cert[1]);
        norms(w2, bb2, h, h, cert[2], cert[3]);
        norms(w3, bb3, 8, h, cert[4], cert[5]);
        bool finite = true;
        for (double v : cert) finite = finite && std::isfinite(v);
        // EPI_MLP_CERTIFY=0 switches the screen off; =all sends every state to the exact evaluation (tests)
        const char* env = getenv("EPI_MLP_CERTIFY");
        certify = finite && !(env && env[0] == '0');
        cert_scale = (env && std::strcmp(env, "all") == 0) ? 1e300 : 1.0;
        return EPI_OK;
    }
    int certify = 0;
    double cert[6] = {0, 0, 0, 0, 0, 0};
    double cert_scale = 1.0;
    DevBuf screen_stats;
    int32_t read_stats(int64_t* screened, int64_t* flagged) const {
        unsigned long long h[2] = {0, 0};
        if (screen_stats.p) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipMemcpy(h, screen_stats.p, sizeof h, hipMemcpyDeviceToHost));
        }
        if (screened) *screened = (int64_t)h[0];
        if (flagged) *flagged = (int64_t)h[1];
        return EPI_OK;
    }
    void fill(MlpArgs& a) const {
        a.nl = nl;
        a.nets = net.as<double>();
        a.net_stride = 0;
        a.per_country = 0;
        a.clip_q = clip_q;
        a.clip_ptr = nullptr;
        a.clip_stride = 0;
        a.certify = certify;
        for (int k = 0; k < 6; ++k) a.cert[k] = cert[k];
        a.cert_scale = cert_scale;
        a.cert_stats = screen_stats.as<unsigned long long>();
    }
};

const char* status_text(int code) {
    switch (code) {
        case EPI_ERR_LOCAL_MASS: return "requirement failed: mass not conserved (local)";
        case EPI_ERR_WORLD_MASS: return "requirement failed: global mass changed during mobility";
        case EPI_ERR_INT_RANGE: return "a compartment left the 32-bit Int range the reference computes in";
        case EPI_ERR_UNSUPPORTED: return "more units left over than positive fractions in a sparse row";
        default: return "device-side failure";
    }
}

// First failing episode of a run, from the device error table.
// `graph` (optional) turns the engine's patch numbers back into the caller's.
int32_t collect_errors(const int32_t* err, int count, int g_begin, int g_stride = 1, const GraphTables* graph = nullptr) {
    std::vector<int32_t> h((size_t)count * 4);
    HIP_TRY(hipMemcpy(h.data(), err, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int e = 0; e < count; ++e)
        if (h[(size_t)e * 4] != 0)
            return fail(h[(size_t)e * 4], "%s (instance %d, country %d, t=%d)", status_text(h[(size_t)e * 4]), g_begin + e * g_stride,
                        graph ? graph->outer(h[(size_t)e * 4 + 1]) : h[(size_t)e * 4 + 1], h[(size_t)e * 4 + 2]);
    return EPI_OK;
}

int32_t check_base_state(const epi_state& b, const char* who) {
    if (!is_count(b.r) || !is_count(b.d) || !is_count(b.v))
        return fail(EPI_ERR_BAD_ARGUMENT, "%s: base r/d/v must be non-negative integers below 2^31", who);
    if (!std::isfinite(b.hosp_cap)) return fail(EPI_ERR_BAD_ARGUMENT, "%s: base hospCap must be finite", who);
    return EPI_OK;
}

// Validation.assertState on the initial states (WorldToy.scala:53): s, i >= 0 by construction;
// round(total) must equal pop, i.e. r + d + v must be 0; 0 <= t <= tMax.
int32_t assert_initial(const epi_state& b) {
    if (b.r + b.d + b.v != 0.0)
        return fail(EPI_ERR_ASSERT_STATE, "requirement failed: mass not conserved: initial r+d+v = %g must be 0", b.r + b.d + b.v);
    if (!(b.t >= 0 && b.t <= b.t_max))
        return fail(EPI_ERR_ASSERT_STATE, "requirement failed: t out of bounds: %d not in [0, %d]", b.t, b.t_max);
    return EPI_OK;
}

}  // namespace

// ================================================================================================
// handles
// ================================================================================================
struct epi_grid {
    int device = 0;
    int n = 0, npad = 0, steps = 0, n_beta = 0, n_travel = 0, seed_idx = 0, seed_deg = 0, seed_long = 0, seed_block = -1;
    epi_state base{};
    GraphTables graph;
    DevBuf qseed, sumw_seed, qtop_seed, beta0, cst, actions, noise;
    DevBuf t_arrive, has_arrive, t_peak, has_peak;
    MlpDevice mlp;
    std::vector<int> greedy_steps;
    EpisodeBuffers work;
    DevBuf loss_tmp, arrival_tmp, peak_tmp;
    int last_count = 0, last_g_begin = 0, last_g_stride = 1;
    const int32_t* last_err = nullptr;  // error table of the last enqueued run (the handle's or the fixed work area's)
    bool pending = false;
    // optional timing of the episode kernel launches (HIP events on the launch stream)
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> open_events;
    double kernel_ms = 0.0;
    long long kernel_launches = 0;
    long long kernel_country_steps = 0;
    DevBuf stats;  // search statistics, allocated at creation when EPI_STATS is set in the environment
    // epi_grid_run_noalloc: the per-episode state lives in the per-GPU fixed work area, which every handle on that
    // GPU shares -- the error table does not: it is this handle's own (allocated at creation, sized for the most
    // episodes one such call can take), so another handle's later call cannot overwrite what status() will read
    DevBuf noalloc_err;
    int noalloc_max_episodes = 0;
    // rollouts whose agents learn before an argmax-Q step (WorldToy.scala:147-167): one agent per (episode, country)
    bool learning = false;
    epi_hyper hyper{};
    epi_learn_hyper learn_hp{};
    std::vector<int32_t> intensity;      // [n] gradAccumSteps of each country's learn call
    DevBuf rwd;                          // [npad] RewardConsts (the agents observe the reward, WorldToy.scala:99)
    DevBuf prev_st, prev_cap, prev_tcur, loc_s, loc_i, reward_dev, net_index;
};

struct epi_world {
    int device = 0;
    int n = 0, npad = 0, batch = 0, seed_idx = 0, validate = 0;
    epi_state base{};
    GraphTables graph;
    DevBuf beta0, cst, rwd;
    MlpDevice mlp;
    bool mlp_set = false;
    EpisodeBuffers work;
    DevBuf act_dev, gauss_dev, reward_dev, flow_s, flow_i, q_tmp;
    DevBuf noise_tab, series_dev, arr_out, peak_out;   // greedy rollouts: per-step noise table, I-series, compact metrics
    // transitions kept for epi_agents_observe_world: the states before the last step and S, I right after env.step
    bool keep_tr = false, have_tr = false;
    DevBuf prev_st, prev_cap, prev_tcur, loc_s, loc_i;
    // compact [batch][n] images of the caller's arrays in page-locked, device-mapped host memory
    PinBuf pin_act, pin_gauss, pin_states, pin_packed, pin_rewards, pin_actions, pin_err;

    std::vector<JavaRandom> env_rnd;  // one stream per country (env.scala:42)
    std::vector<int32_t> env_seeds;
    bool have_flows = false;
    // optional timing of the episode kernel (HIP events on the launch stream)
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double kernel_ms = 0.0;
    long long kernel_launches = 0, kernel_country_steps = 0;
    ~epi_world() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
};

// ================================================================================================
// basics
// ================================================================================================
EPI_API int32_t epi_version(void) { return 100; }

EPI_API int32_t epi_last_error(char* buf, size_t cap) {
    if (buf && cap > 0) {
        const size_t k = g_last_error.size() < cap - 1 ? g_last_error.size() : cap - 1;
        std::memcpy(buf, g_last_error.data(), k);
        buf[k] = 0;
    }
    return (int32_t)g_last_error.size();
}

EPI_API int32_t epi_last_status(void) { return g_last_code; }

EPI_API int32_t epi_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

// ================================================================================================
// host-side JVM-compatible streams
// ================================================================================================
EPI_API int32_t epi_java_gaussians(int64_t seed, int32_t n, double* out) {
    if (n < 0 || (n > 0 && !out)) return fail(EPI_ERR_BAD_ARGUMENT, "epi_java_gaussians: bad arguments");
    JavaRandom r(seed);
    for (int k = 0; k < n; ++k) out[k] = r.nextGaussian();
    return EPI_OK;
}

EPI_API int32_t epi_java_gaussian_table(int64_t seed0, int32_t streams, int32_t draws, double* out, int32_t device) {
    if (streams < 0 || draws < 0 || ((size_t)streams * draws > 0 && !out))
        return fail(EPI_ERR_BAD_ARGUMENT, "epi_java_gaussian_table: bad arguments");
    if ((size_t)streams * draws == 0) return EPI_OK;
    int32_t rc = use_device(device);
    if (rc) return rc;
    DevBuf tab;
    HIP_TRY(tab.ensure(sizeof(double) * (size_t)streams * draws));
    HIP_TRY(launch_gaussian_table(tab.as<double>(), nullptr, streams, draws, (size_t)streams, seed0, nullptr));
    HIP_TRY(hipMemcpy(out, tab.p, sizeof(double) * (size_t)streams * draws, hipMemcpyDeviceToHost));
    return EPI_OK;
}

EPI_API int32_t epi_agent_actions(const epi_hyper* hp, int32_t steps, int32_t* out) {
    if (!hp || steps < 0 || (steps > 0 && !out)) return fail(EPI_ERR_BAD_ARGUMENT, "epi_agent_actions: bad arguments");
    AgentStream ag;
    const Hyper h = to_hyper(*hp);
    for (int t = 0; t < steps; ++t) out[t] = ag.act(h);
    return EPI_OK;
}

EPI_API int32_t epi_mlp_init(int32_t n_in, int32_t hidden, int32_t n_out, double* W1, double* W2, double* W3) {
    if (n_in < 1 || hidden < 1 || n_out < 1 || !W1 || !W2 || !W3) return fail(EPI_ERR_BAD_ARGUMENT, "epi_mlp_init: bad arguments");
    std::vector<double> a, b, c;
    mlp_init(n_in, hidden, n_out, a, b, c);
    std::memcpy(W1, a.data(), a.size() * sizeof(double));
    std::memcpy(W2, b.data(), b.size() * sizeof(double));
    std::memcpy(W3, c.data(), c.size() * sizeof(double));
    return EPI_OK;
}

EPI_API int32_t epi_scale_row(const double* W, int32_t n, int32_t row, double factor, double* out) {
    if (!W || !out || n < 1 || row < 0 || row >= n) return fail(EPI_ERR_BAD_ARGUMENT, "epi_scale_row: bad arguments");
    if (out != W) std::memcpy(out, W, sizeof(double) * (size_t)n * n);
    scale_row_inplace(out + (size_t)row * n, n, row, factor);
    return EPI_OK;
}

EPI_API int32_t epi_grid_argmin(const double* loss, int32_t count, int32_t* best_idx, double* best_loss) {
    if (count < 0 || (count > 0 && !loss)) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_argmin: bad arguments");
    double bl = std::numeric_limits<double>::max();  // Calibrate.scala:96
    int32_t bi = 0;
    for (int g = 0; g < count; ++g)
        if (loss[g] < bl) { bl = loss[g]; bi = g; }  // :112
    if (best_idx) *best_idx = bi;
    if (best_loss) *best_loss = bl;
    return EPI_OK;
}

// ================================================================================================
// batched unit operations
// ================================================================================================
// The unit operations below (allocateInt, env.step, actGreedy for caller-held arrays) are called once per
// reference call by a shim that forwards EpidemicEnv.allocateInt or env.step one at a time: their device staging
// is kept per device for the life of the process and only grows, so a call allocates nothing once the sizes have
// been seen.  One caller at a time per device (a mutex serialises the rest).
namespace {
struct UnitScratch {
    std::mutex mu;
    DevBuf b[8];
    MlpDevice mlp;                 // network of the last epi_mlp_greedy_batch call ...
    std::vector<double> mlp_key;   // ... and the weights it was built from (re-uploaded only when they change)
};
UnitScratch* unit_scratch(int device) {
    static std::mutex mu;
    static UnitScratch* per_device[kMaxScratchDevices];   // never freed: it must outlive every caller
    if (device < 0 || device >= kMaxScratchDevices) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!per_device[device]) per_device[device] = new UnitScratch();
    return per_device[device];
}
thread_local int g_unit_device = 0;
}  // namespace

EPI_API int32_t epi_set_device(int32_t device) {
    int32_t rc = use_device(device);
    if (rc) return rc;
    if (device >= kMaxScratchDevices) return fail(EPI_ERR_UNSUPPORTED, "epi_set_device: devices 0..%d", kMaxScratchDevices - 1);
    g_unit_device = device;
    return EPI_OK;
}

EPI_API int32_t epi_alloc_int_batch(const int32_t* totals, const double* w, int32_t n, int32_t batch, int32_t* out) {
    if (n < 0 || batch < 0) return fail(EPI_ERR_BAD_ARGUMENT, "epi_alloc_int_batch: negative size");
    if (n == 0 || batch == 0) return EPI_OK;
    if (!totals || !w || !out) return fail(EPI_ERR_BAD_ARGUMENT, "epi_alloc_int_batch: null pointer");
    if (n > kMaxDenseN) return fail(EPI_ERR_BAD_ARGUMENT, "epi_alloc_int_batch: n = %d exceeds the %d weights one wave holds", n, kMaxDenseN);
    const size_t cells = (size_t)n * batch;
    for (size_t k = 0; k < cells; ++k)
        if (!std::isfinite(w[k]) || w[k] < 0.0) return fail(EPI_ERR_BAD_ARGUMENT, "epi_alloc_int_batch: weight %zu = %g must be finite and >= 0", k, w[k]);
    int32_t rc = use_device(g_unit_device);
    if (rc) return rc;
    UnitScratch* us = unit_scratch(g_unit_device);
    std::lock_guard<std::mutex> lock(us->mu);
    DevBuf &dt = us->b[0], &dw = us->b[1], &dout = us->b[2];
    HIP_TRY(dt.ensure(sizeof(int32_t) * (size_t)batch));
    HIP_TRY(dw.ensure(sizeof(double) * cells));
    HIP_TRY(dout.ensure(sizeof(int32_t) * cells));
    HIP_TRY(hipMemcpyAsync(dt.p, totals, sizeof(int32_t) * (size_t)batch, hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(dw.p, w, sizeof(double) * cells, hipMemcpyHostToDevice, nullptr));
    AllocBatchArgs a{n, dense_slots_for(n) * 64, batch, dt.as<int32_t>(), dw.as<double>(), dout.as<int32_t>()};
    HIP_TRY(launch_alloc_batch(a, nullptr));
    HIP_TRY(hipMemcpy(out, dout.p, sizeof(int32_t) * cells, hipMemcpyDeviceToHost));   // waits for the launch
    return EPI_OK;
}

EPI_API int32_t epi_alloc_int(int32_t total, const double* w, int32_t n, int32_t* out) {
    return epi_alloc_int_batch(&total, w, n, 1, out);
}

EPI_API int32_t epi_env_step_batch(const epi_env_params* env, const epi_profile* prof, const epi_state* in,
                                   const int32_t* action, const double* gauss, epi_state* out, double* reward,
                                   uint8_t* done, int32_t batch) {
    if (batch < 0) return fail(EPI_ERR_BAD_ARGUMENT, "epi_env_step_batch: negative batch");
    if (batch == 0) return EPI_OK;
    if (!env || !prof || !in || !action || !gauss || !out) return fail(EPI_ERR_BAD_ARGUMENT, "epi_env_step_batch: null pointer");
    for (int b = 0; b < batch; ++b) {
        const epi_state& s = in[b];
        if (!std::isfinite(s.s) || !std::isfinite(s.i) || !std::isfinite(s.r) || !std::isfinite(s.d) || !std::isfinite(s.v) ||
            !std::isfinite(s.hosp_cap) || !std::isfinite(gauss[b]))
            return fail(EPI_ERR_BAD_ARGUMENT, "epi_env_step_batch: row %d holds a non-finite value", b);
    }
    int32_t rc = use_device(g_unit_device);
    if (rc) return rc;
    UnitScratch* us = unit_scratch(g_unit_device);
    std::lock_guard<std::mutex> lock(us->mu);
    DevBuf &de = us->b[0], &dp = us->b[1], &di = us->b[2], &da = us->b[3], &dg = us->b[4], &dout = us->b[5], &dr = us->b[6], &dd = us->b[7];
    const size_t B = (size_t)batch;
    HIP_TRY(de.ensure(B * sizeof(epi_env_params)));
    HIP_TRY(dp.ensure(B * sizeof(epi_profile)));
    HIP_TRY(di.ensure(B * sizeof(epi_state)));
    HIP_TRY(da.ensure(B * sizeof(int32_t)));
    HIP_TRY(dg.ensure(B * sizeof(double)));
    HIP_TRY(dout.ensure(B * sizeof(epi_state)));
    HIP_TRY(dr.ensure(B * sizeof(double)));
    HIP_TRY(dd.ensure(B));
    HIP_TRY(hipMemcpyAsync(de.p, env, B * sizeof(epi_env_params), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(dp.p, prof, B * sizeof(epi_profile), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(di.p, in, B * sizeof(epi_state), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(da.p, action, B * sizeof(int32_t), hipMemcpyHostToDevice, nullptr));
    HIP_TRY(hipMemcpyAsync(dg.p, gauss, B * sizeof(double), hipMemcpyHostToDevice, nullptr));
    EnvBatchArgs a{batch, de.as<epi_env_params>(), dp.as<epi_profile>(), di.as<epi_state>(), da.as<int32_t>(),
                   dg.as<double>(), dout.as<epi_state>(), dr.as<double>(), dd.as<uint8_t>()};
    HIP_TRY(launch_env_batch(a, nullptr));
    HIP_TRY(hipMemcpy(out, dout.p, B * sizeof(epi_state), hipMemcpyDeviceToHost));   // waits for the launch
    if (reward) HIP_TRY(hipMemcpy(reward, dr.p, B * sizeof(double), hipMemcpyDeviceToHost));
    if (done) HIP_TRY(hipMemcpy(done, dd.p, B, hipMemcpyDeviceToHost));
    return EPI_OK;
}

EPI_API int32_t epi_mlp_greedy_batch(const epi_state* states, int32_t batch, int32_t hidden, const double* W1,
                                     const double* b1, const double* W2, const double* b2, const double* W3,
                                     const double* b3, double clip_q, double* q, int32_t* action) {
    if (batch < 0) return fail(EPI_ERR_BAD_ARGUMENT, "epi_mlp_greedy_batch: negative batch");
    if (batch == 0) return EPI_OK;
    if (!states || !W1 || !W2 || !W3) return fail(EPI_ERR_BAD_ARGUMENT, "epi_mlp_greedy_batch: null pointer");
    if (hidden < 1 || hidden > 256) return fail(EPI_ERR_UNSUPPORTED, "hidden = %d: the device network supports 1..256 units", hidden);
    int32_t rc = use_device(g_unit_device);
    if (rc) return rc;
    UnitScratch* us = unit_scratch(g_unit_device);
    std::lock_guard<std::mutex> lock(us->mu);
    {   // the network is uploaded again only when hidden, clip or any weight differs from the last call's
        const size_t h = (size_t)hidden;
        std::vector<double> key;
        key.reserve(2 + h * 7 + h * h + 8 * h + 2 * h + 8 + 3);
        key.push_back((double)hidden); key.push_back(clip_q);
        key.insert(key.end(), W1, W1 + h * 7); key.insert(key.end(), W2, W2 + h * h); key.insert(key.end(), W3, W3 + 8 * h);
        key.push_back(b1 ? 1.0 : 0.0); if (b1) key.insert(key.end(), b1, b1 + h);
        key.push_back(b2 ? 1.0 : 0.0); if (b2) key.insert(key.end(), b2, b2 + h);
        key.push_back(b3 ? 1.0 : 0.0); if (b3) key.insert(key.end(), b3, b3 + 8);
        if (key.size() != us->mlp_key.size() || std::memcmp(key.data(), us->mlp_key.data(), key.size() * sizeof(double)) != 0) {
            rc = us->mlp.upload(hidden, W1, b1, W2, b2, W3, b3, clip_q);
            if (rc) { us->mlp_key.clear(); return rc; }
            us->mlp_key.swap(key);
        }
    }
    MlpDevice& m = us->mlp;
    DevBuf &ds = us->b[0], &dq = us->b[1], &da = us->b[2];
    HIP_TRY(ds.ensure((size_t)batch * sizeof(epi_state)));
    HIP_TRY(dq.ensure((size_t)batch * 8 * sizeof(double)));
    HIP_TRY(da.ensure((size_t)batch * sizeof(int32_t)));
    HIP_TRY(hipMemcpyAsync(ds.p, states, (size_t)batch * sizeof(epi_state), hipMemcpyHostToDevice, nullptr));
    MlpArgs a{};
    a.rows = batch; a.n = 1; a.npad = 1; a.count = batch;
    a.states = ds.as<epi_state>();
    m.fill(a);
    a.act_out = da.as<int32_t>();
    a.q_out = dq.as<double>();
    HIP_TRY(launch_mlp(a, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    if (q) HIP_TRY(hipMemcpy(q, dq.p, (size_t)batch * 8 * sizeof(double), hipMemcpyDeviceToHost));
    if (action) HIP_TRY(hipMemcpy(action, da.p, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToHost));
    return EPI_OK;
}

// ================================================================================================
// calibration sweep
// ================================================================================================
static int32_t grid_build(epi_grid* G, const epi_grid_desc* d) {
    if (d->n < 1) return fail(EPI_ERR_BAD_ARGUMENT, "grid: n = %d", d->n);
    if (!d->pop || d->seed_idx < 0 || d->seed_idx >= d->n || d->steps < 0 || d->n_beta < 0 || d->n_travel < 0)
        return fail(EPI_ERR_BAD_ARGUMENT, "grid: bad sizes or null pop");
    if ((d->n_beta > 0 && !d->betas) || (d->n_travel > 0 && !d->travel)) return fail(EPI_ERR_BAD_ARGUMENT, "grid: null grid axis");
    if (d->heterogeneous && !d->prof) return fail(EPI_ERR_BAD_ARGUMENT, "grid: heterogeneous sweep needs country profiles");
    int32_t rc = check_base_state(d->base, "grid");
    if (rc) return rc;
    rc = assert_initial(d->base);  // validate = true is hard-wired in Calibrate.rollout (Calibrate.scala:64)
    if (rc) return rc;
    rc = use_device(d->device);
    if (rc) return rc;
    // EPI_CREATE_TIMERS=1: wall time of every stage of the build on stderr (diagnostic)
    const bool timers = getenv("EPI_CREATE_TIMERS") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (!timers) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[epi create] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    G->device = d->device;
    G->n = d->n; G->steps = d->steps; G->n_beta = d->n_beta; G->n_travel = d->n_travel; G->seed_idx = d->seed_idx;
    G->base = d->base;
    G->hyper = d->hp;
    // a sparse graph is renumbered inside the engine (GraphTables::renumbered): every per-country table below is
    // filed under in(c), and the loss kernel reads arrival / peak back through the same map
    rc = G->graph.build(d->n, d->pop, d->conn, d->csr_rowptr, d->csr_col, d->csr_val, true);
    if (rc) return rc;
    stamp("graph tables");
    const int n = d->n, npad = G->graph.npad;
    G->npad = npad;
    auto in = [&](int c) { return (size_t)G->graph.inner(c); };
    G->seed_idx = (int)in(d->seed_idx);

    // seed-row variants: ToyConn.scaleRow(W, seedIdx, ts) touches only that row (Calibrate.scala:29)
    if (G->graph.sparse) {
        // the stored entries of the row in column order: scaling and the row-sum cap of scaleRow add
        // and multiply the zeros of the other columns without changing anything
        const int rs = G->graph.h_rowptr[d->seed_idx], deg = G->graph.h_rowptr[d->seed_idx + 1] - rs;
        const int nt = d->n_travel > 0 ? d->n_travel : 1;
        const bool seed_long = deg > kLaneSlots;
        // a long seed row is read by one wave as whole groups of 64 padded slots; otherwise the whole 64-row block of the
        // seed row is copied once per travel scale (kLaneSlots slots each) with that one row replaced
        const int slots = seed_long ? (int)long_row_slots(deg) : kLaneSlots;
        const int spos = G->graph.h_pos[d->seed_idx];
        const int sb = seed_long ? 0 : spos / 64, sl = seed_long ? 0 : spos % 64;
        G->seed_block = seed_long ? -1 : sb;
        G->seed_long = seed_long ? 1 : 0;
        std::vector<double> hq((size_t)nt * slots * (seed_long ? 1 : 64), 0.0), hs((size_t)nt, 0.0), row((size_t)(deg > 0 ? deg : 1));
        std::vector<double> qrow((size_t)std::max(64, slots), 0.0);
        std::vector<double> seed_block_q;
        if (!seed_long) {
            rc = G->graph.download_block_q(sb, seed_block_q);
            if (rc) return rc;
        }
        for (int k = 0; k < d->n_travel; ++k) {
            if (!std::isfinite(d->travel[k]) || d->travel[k] < 0.0) return fail(EPI_ERR_BAD_ARGUMENT, "grid: travel scale %d = %g", k, d->travel[k]);
            double s = 0.0;
            for (int e = 0; e < deg; ++e) {
                row[e] = G->graph.h_val[rs + e];
                if (G->graph.h_col[rs + e] != d->seed_idx) { row[e] *= d->travel[k]; s += row[e]; }
            }
            const double cap = std::fmin(0.2, s);
            if (s > cap && s > 0) {
                const double sc = cap / s;
                for (int e = 0; e < deg; ++e)
                    if (G->graph.h_col[rs + e] != d->seed_idx) row[e] *= sc;
            }
            std::fill(qrow.begin(), qrow.end(), 0.0);
            rc = GraphTables::normalise_sparse_row(row.data(), G->graph.h_col.data() + rs, deg, d->seed_idx, qrow.data(), &hs[k]);
            if (rc) return rc;
            if (seed_long) {
                std::copy(qrow.begin(), qrow.begin() + slots, hq.begin() + (size_t)k * slots);
            } else {
                double* blk = &hq[(size_t)k * slots * 64];
                std::copy(seed_block_q.begin(), seed_block_q.end(), blk);   // whole slot groups of the block, the rest stays 0
                for (int e = 0; e < slots; ++e) blk[(size_t)e * 64 + sl] = e < deg ? qrow[e] : 0.0;
            }
        }
        G->seed_deg = slots;
        HIP_TRY(G->qseed.upload(hq));
        HIP_TRY(G->sumw_seed.upload(hs));
    } else {
        // ToyConn.scaleRow(W, seedIdx, ts) for every travel scale, its normalised row, row sum and top weights: one
        // workgroup per scale on the device, from the matrix the graph tables were built from
        const int nt = d->n_travel > 0 ? d->n_travel : 1;
        for (int k = 0; k < d->n_travel; ++k)
            if (!std::isfinite(d->travel[k]) || d->travel[k] < 0.0) return fail(EPI_ERR_BAD_ARGUMENT, "grid: travel scale %d = %g", k, d->travel[k]);
        HIP_TRY(G->qseed.ensure(sizeof(double) * (size_t)nt * npad));
        HIP_TRY(G->qtop_seed.ensure(sizeof(double) * (size_t)nt * kTopWeights));
        HIP_TRY(G->sumw_seed.ensure(sizeof(double) * (size_t)nt));
        if (d->n_travel > 0) {
            DevBuf d_travel;
            HIP_TRY(d_travel.upload(std::vector<double>(d->travel, d->travel + d->n_travel)));
            PrepareRowsArgs pa{};
            pa.n = n; pa.npad = npad; pa.W = G->graph.d_W.as<double>(); pa.n_rows = d->n_travel; pa.seed_idx = d->seed_idx;
            pa.travel = d_travel.as<double>();
            pa.q = G->qseed.as<double>(); pa.sumw = G->sumw_seed.as<double>(); pa.qtop = G->qtop_seed.as<double>();
            HIP_TRY(launch_prepare_rows(pa, nullptr));
            HIP_TRY(hipDeviceSynchronize());   // d_travel goes out of scope
        }
    }
    stamp("seed-row variants");
    G->graph.d_W.release();   // the tables are complete: the matrix itself is not read again
    G->graph.W.clear(); G->graph.W.shrink_to_fit();
    // per-country environments (Calibrate.scala:31-51)
    const Hyper hp = to_hyper(d->hp);
    {
        const int nb = d->n_beta > 0 ? d->n_beta : 1;
        std::vector<double> hb((size_t)nb * npad, 0.0), hc((size_t)CST_COUNT * npad, 0.0);
        for (int c = 0; c < n; ++c) {
            const epi_profile& p = d->heterogeneous ? d->prof[c] : kDefaultProfile;
            epi_env_params e;
            e.base_gamma = 0.12;
            e.base_ifr = d->heterogeneous ? 0.008 * (1.0 + p.age_elderly * 2.0) : 0.008;
            e.noise = d->heterogeneous ? 0.05 * (1.0 + (1.0 - p.econ_resilience)) : 0.05;
            e.reward_clip = hp.reward_clip;
            e.risk_factor = 0.0;
            for (int b = 0; b < d->n_beta; ++b) {
                const double beta = d->betas[b];
                if (!std::isfinite(beta)) return fail(EPI_ERR_BAD_ARGUMENT, "grid: beta %d is not finite", b);
                e.base_beta = d->heterogeneous
                                  ? beta * (1.0 + (p.population_density / 1000.0) * 0.5) * (3.0 - p.healthcare_capacity * 2.0)
                                  : beta;
                hb[(size_t)b * npad + in(c)] = make_step_consts(e, p).beta0;
            }
            e.base_beta = 0.0;
            const StepConsts sc = make_step_consts(e, p);
            hc[(size_t)CST_GAMMA * npad + in(c)] = sc.gamma;
            hc[(size_t)CST_IFR0 * npad + in(c)] = sc.ifr0;
            hc[(size_t)CST_CNOISE * npad + in(c)] = sc.cnoise;
            hc[(size_t)CST_EFF1 * npad + in(c)] = sc.eff1;
            hc[(size_t)CST_EFF2 * npad + in(c)] = sc.eff2;
            hc[(size_t)CST_EFF3 * npad + in(c)] = sc.eff3;
            hc[(size_t)CST_SURGE * npad + in(c)] = sc.surge;
            hc[(size_t)CST_VAX6 * npad + in(c)] = sc.vax6;
            hc[(size_t)CST_VAX7 * npad + in(c)] = sc.vax7;
        }
        HIP_TRY(G->beta0.upload(hb));
        HIP_TRY(G->cst.upload(hc));
    }
    stamp("env constants");
    // noise draws z[t][c] (env.scala:91) and the shared action stream (DDQNAgent.scala:26-32):
    // every rollout builds fresh envs and agents, so these are the same for all grid points.
    {
        const int T = d->steps > 0 ? d->steps : 1;
        if (d->heterogeneous) {
            // one stream per country, seeded idx + 42 (Calibrate.scala:36): drawn on the device, a thread per stream
            HIP_TRY(G->noise.ensure(sizeof(double) * (size_t)T * npad));
            HIP_TRY(hipMemsetAsync(G->noise.p, 0, sizeof(double) * (size_t)T * npad, nullptr));
            HIP_TRY(launch_gaussian_table(G->noise.as<double>(), G->graph.renumbered ? G->graph.d_id_of.as<int32_t>() : nullptr, n,
                                          d->steps, (size_t)npad, 42, nullptr));
        } else {
            std::vector<double> hz((size_t)T * npad, 0.0);
            JavaRandom r(7);  // Calibrate.scala:32: every env is seeded 7
            for (int t = 0; t < d->steps; ++t) {
                const double z = r.nextGaussian();
                for (int c = 0; c < n; ++c) hz[(size_t)t * npad + c] = z;
            }
            HIP_TRY(G->noise.upload(hz));
        }
        std::vector<int8_t> ha((size_t)T * npad, 0);
        AgentStream ag;  // all agents share seed 9 and the hyper-parameters: one stream serves all
        G->greedy_steps.clear();
        for (int t = 0; t < d->steps; ++t) {
            const int act = ag.act(hp);
            if (act == kGreedyAction) G->greedy_steps.push_back(t);
            for (int c = 0; c < n; ++c) ha[(size_t)t * npad + c] = (int8_t)(act == kGreedyAction ? 0 : act);
        }
        HIP_TRY(G->actions.upload(ha));
        // WorldToy.stepOne lets every agent learn once gStep >= learnEvery and its replay buffer holds 4 * batchSize
        // transitions (WorldToy.scala:150-167; one transition per step, :99).  An argmax-Q step at or after that point
        // consults trained weights: such a sweep runs with one learning agent per (episode, country), step by step.
        const long long learn_from = std::max<long long>(1, std::max<long long>(hp.learn_every, 4LL * hp.batch_size));
        const bool ever_learns = !d->learn || (long long)d->learn->replay_capacity >= 4LL * hp.batch_size;   // a smaller buffer never fills
        int first_trained = -1;
        for (int gs : G->greedy_steps)
            if (gs >= learn_from && ever_learns) { first_trained = gs; break; }
        if (first_trained >= 0) {
            if (!d->learn)
                return fail(EPI_ERR_UNSUPPORTED,
                            "grid: the agents start learning after step %lld (learnEvery %d, batchSize %d) and step %d takes argmax Q: "
                            "rollouts that consult trained weights need epi_grid_desc.learn (the HyperParams of DDQNAgent.learn)",
                            learn_from, hp.learn_every, hp.batch_size, first_trained);
            if (G->graph.sparse)
                return fail(EPI_ERR_UNSUPPORTED, "grid: rollouts whose agents learn (from step %lld on) are run for dense worlds of at most "
                                                 "1,024 countries; this one has %d patches", learn_from, n);
            if (d->learn->hidden != hp.hidden)
                return fail(EPI_ERR_BAD_ARGUMENT, "grid: learn->hidden = %d but hp.hidden = %d (one HyperParams record, config.scala:13)",
                            d->learn->hidden, hp.hidden);
            if (d->learn->replay_capacity <= 0 || d->learn->target_update_every <= 0 || hp.batch_size <= 0 || hp.learn_every < 0)
                return fail(EPI_ERR_BAD_ARGUMENT, "grid: learn: replay_capacity, target_update_every and batch_size must be positive");
            G->learning = true;
            G->learn_hp = *d->learn;
            G->intensity.assign((size_t)n, 1);
            for (int c = 0; c < n && d->learn_intensity; ++c) {
                if (d->learn_intensity[c] < 1) return fail(EPI_ERR_BAD_ARGUMENT, "grid: learn_intensity[%d] = %d", c, d->learn_intensity[c]);
                G->intensity[(size_t)c] = d->learn_intensity[c];
            }
            std::vector<RewardConsts> hr((size_t)npad);
            std::memset(hr.data(), 0, hr.size() * sizeof(RewardConsts));
            for (int c = 0; c < n; ++c) {
                epi_env_params e{};
                e.reward_clip = hp.reward_clip;   // Calibrate.scala:32,47; riskFactor keeps its default 0.0
                e.risk_factor = 0.0;
                hr[(size_t)c] = make_reward_consts(e, d->heterogeneous ? d->prof[c] : kDefaultProfile);
            }
            HIP_TRY(G->rwd.upload(hr));
        }
    }
    stamp("noise and actions");
    if (!G->greedy_steps.empty()) {
        std::vector<double> w1, w2, w3;
        mlp_init(7, hp.hidden, 8, w1, w2, w3);  // untrained online network (DDQNAgent.scala:8)
        rc = G->mlp.upload(hp.hidden, w1.data(), nullptr, w2.data(), nullptr, w3.data(), nullptr, hp.clip_q);
        if (rc) return rc;
    }
    // targets (Calibrate.Targets maps as value + presence arrays)
    {
        std::vector<double> ta((size_t)n, 0.0), tp((size_t)n, 0.0);
        std::vector<uint8_t> ha((size_t)n, 0), hk((size_t)n, 0);
        for (int c = 0; c < n; ++c) {
            if (d->has_arrive && d->has_arrive[c]) { ha[c] = 1; ta[c] = d->t_arrive[c]; }
            if (d->has_peak && d->has_peak[c]) { hk[c] = 1; tp[c] = d->t_peak[c]; }
        }
        HIP_TRY(G->t_arrive.upload(ta));
        HIP_TRY(G->t_peak.upload(tp));
        HIP_TRY(G->has_arrive.upload(ha));
        HIP_TRY(G->has_peak.upload(hk));
    }
    stamp("network and targets");
    // everything a later call must not do: kernel attributes of this device, the statistics block, the error table of
    // the allocation-free entry point
    HIP_TRY(prepare_device_kernels());
    if (getenv("EPI_STATS")) {
        HIP_TRY(G->stats.ensure(64 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(G->stats.p, 0, 64 * sizeof(unsigned long long)));
    }
    {
        const size_t per_episode = scratch_need(1, (size_t)npad, G->graph.sparse);
        const size_t fit = kNoallocScratchBytes / per_episode;   // an upper bound: sections are padded per call, not per episode
        const size_t grid_points = (size_t)std::max(1, d->n_beta * d->n_travel);
        G->noalloc_max_episodes = (int)std::min(fit, grid_points);
        if (G->noalloc_max_episodes > 0) {
            HIP_TRY(G->noalloc_err.ensure((size_t)G->noalloc_max_episodes * 4 * sizeof(int32_t)));
            HIP_TRY(hipMemset(G->noalloc_err.p, 0, (size_t)G->noalloc_max_episodes * 4 * sizeof(int32_t)));
        }
    }
    return EPI_OK;
}

EPI_API epi_grid* epi_grid_create(const epi_grid_desc* desc) {
    if (!desc) { fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_create: null descriptor"); return nullptr; }
    std::unique_ptr<epi_grid> G(new epi_grid());
    if (grid_build(G.get(), desc) != EPI_OK) return nullptr;
    return G.release();
}

static int32_t drain_events(epi_grid* g);

EPI_API int32_t epi_grid_destroy(epi_grid* g) {
    if (!g) return EPI_OK;
    (void)hipSetDevice(g->device);
    (void)drain_events(g);
    if (g->stats.p) {
        unsigned long long h[64];
        if (hipMemcpy(h, g->stats.p, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
            for (int k = 0; k < 2; ++k) {  // threshold-search statistics of the dense kernel
                const unsigned long long* s = h + 32 * k;
                if (s[0] == 0) continue;
                fprintf(stderr, "[epi stats %s] rows %llu probes %llu (%.2f/row) cold %llu ties %llu hist:", k ? "I" : "S", s[0], s[1],
                        (double)s[1] / (double)s[0], s[2], s[3]);
                for (int b = 0; b < 28; ++b) fprintf(stderr, " %llu", s[4 + b]);
                fprintf(stderr, "\n");
            }
            // phase timers of the sparse kernel (builds with -DEPI_CSR_TIMERS); they share slots with the histogram above
            if (h[0] == 0 && h[32] == 0 && (h[48] | h[51] | h[53]))
                fprintf(stderr, "[epi cycles] pass1 %llu select %llu pass2 %llu | local+apply %llu barrier %llu mobility %llu barrier %llu | "
                                "block set-up %llu\n", h[48], h[49], h[50], h[51], h[52], h[53], h[54], h[55]);
        }
    }
    delete g;
    return EPI_OK;
}

// Launches one segment of the episode kernel, bracketed by HIP events when timing is on.
static int32_t timed_episode(epi_grid* g, const EpisodeArgs& ea, hipStream_t s) {
    if (!g->timing) {
        HIP_TRY(launch_episode(ea, s));
        return EPI_OK;
    }
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    HIP_TRY(hipEventRecord(a, s));
    HIP_TRY(launch_episode(ea, s));
    HIP_TRY(hipEventRecord(b, s));
    g->open_events.emplace_back(a, b);
    g->kernel_country_steps += (long long)ea.count * ea.n * (ea.t_end - ea.t_begin);
    return EPI_OK;
}

static int32_t drain_events(epi_grid* g) {
    for (auto& ev : g->open_events) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(ev.second));
        HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
        g->kernel_ms += ms;
        g->kernel_launches += 1;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    g->open_events.clear();
    return EPI_OK;
}

EPI_API int32_t epi_grid_set_timing(epi_grid* g, int32_t on) {
    if (!g) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_set_timing: null handle");
    HIP_TRY(hipSetDevice(g->device));
    int32_t rc = drain_events(g);
    if (rc) return rc;
    g->timing = on != 0;
    g->kernel_ms = 0.0;
    g->kernel_launches = 0;
    g->kernel_country_steps = 0;
    return EPI_OK;
}

EPI_API int32_t epi_grid_kernel_time(epi_grid* g, double* total_ms, int64_t* launches, int64_t* country_steps) {
    if (!g) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_kernel_time: null handle");
    HIP_TRY(hipSetDevice(g->device));
    int32_t rc = drain_events(g);
    if (rc) return rc;
    if (total_ms) *total_ms = g->kernel_ms;
    if (launches) *launches = g->kernel_launches;
    if (country_steps) *country_steps = g->kernel_country_steps;
    return EPI_OK;
}

EPI_API int32_t epi_grid_status(epi_grid* g) {
    if (!g) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_status: null handle");
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(hipDeviceSynchronize());
    if (!g->pending) return EPI_OK;
    g->pending = false;
    return collect_errors(g->last_err, g->last_count, g->last_g_begin, g->last_g_stride, &g->graph);
}

EPI_API int32_t epi_grid_run(epi_grid* g, int32_t g_begin, int32_t g_count, double* loss, int32_t* arrival, double* peak,
                             int32_t out_on_device, void* stream) {
    return epi_grid_run_strided(g, g_begin, 1, g_count, loss, arrival, peak, out_on_device, stream);
}

// The episodes of a sweep whose agents learn before an argmax-Q step (WorldToy.stepOne as Calibrate.rollout calls it,
// Calibrate.scala:69-73): every (episode, country) pair is a node with a DDQNAgent of its own (WorldToy.scala:46-65), so
// the episodes advance one step per launch and after every step the agents observe their transition (:99) and learn
// when their schedule says so (:147-167).  All agents share HyperParams and the action stream's seed
// (DDQNAgent.scala:14), so they explore on the same steps and, when they do, draw the same action; on an argmax-Q
// step each asks its own online network.  Blocking: replay sampling is host work (ReplayBuffer.scala:21-22).
static int32_t grid_run_learning(epi_grid* g, EpisodeArgs ea, const WorkPtrs& wp, hipStream_t s) {
    const int n = g->n, npad = g->npad, count = ea.count;
    const size_t cells = (size_t)count * npad;
    const long long n_agents = (long long)count * n;
    if (n_agents > 16384)
        return fail(EPI_ERR_UNSUPPORTED, "grid: %d episodes x %d countries = %lld learning agents in one call, at most 16384: run fewer grid "
                                         "points per call", count, n, n_agents);
    const Hyper hp = to_hyper(g->hyper);
    int max_intensity = 1;
    for (int v : g->intensity) max_intensity = std::max(max_intensity, v);
    epi_learn_hyper lh = g->learn_hp;
    lh.max_need = hp.batch_size * max_intensity;          // DDQNAgent.scala:52
    // a buffer that can hold every transition of the rollout never wraps: no need to allocate beyond the rollout's length
    lh.replay_capacity = std::min(lh.replay_capacity, std::max(1, g->steps));
    std::vector<epi_learn_hyper> lhs((size_t)n_agents, lh);
    struct AgentsOwner {
        epi_agents* p = nullptr;
        ~AgentsOwner() { if (p) epi_agents_destroy(p); }
    } agents;
    agents.p = epi_agents_create((int32_t)n_agents, lhs.data(), g->device);
    if (!agents.p) return g_last_code ? g_last_code : EPI_ERR_DEVICE;
    AgentsNetView view{};
    if (int32_t rc = agents_net_view(agents.p, &view)) return rc;
    HIP_TRY(hipSetDevice(g->device));
    HIP_TRY(g->prev_st.ensure(cells * 5 * sizeof(int32_t)));
    HIP_TRY(g->prev_cap.ensure(cells * sizeof(double)));
    HIP_TRY(g->prev_tcur.ensure(cells * sizeof(int32_t)));
    HIP_TRY(g->loc_s.ensure(cells * sizeof(int32_t)));
    HIP_TRY(g->loc_i.ensure(cells * sizeof(int32_t)));
    HIP_TRY(g->reward_dev.ensure(cells * sizeof(double)));
    {   // network of every (episode, padded country) cell: agent e * n + c; padding columns are evaluated and ignored
        std::vector<int32_t> ni(cells, 0);
        for (int e = 0; e < count; ++e)
            for (int c = 0; c < n; ++c) ni[(size_t)e * npad + c] = e * n + c;
        HIP_TRY(g->net_index.upload(ni));
    }
    ea.rwd = g->rwd.as<RewardConsts>();
    ea.reward_out = g->reward_dev.as<double>();
    ea.loc_s = g->loc_s.as<int32_t>(); ea.loc_i = g->loc_i.as<int32_t>();
    ea.act_episode = wp.act_ep;

    WorldTransitionView tv{};
    tv.device = g->device; tv.n = n; tv.npad = npad; tv.batch = count; tv.t_max = g->base.t_max;
    tv.plane = cells;
    tv.prev_st = g->prev_st.as<int32_t>(); tv.prev_cap = g->prev_cap.as<double>(); tv.prev_tcur = g->prev_tcur.as<int32_t>();
    tv.loc_s = g->loc_s.as<int32_t>(); tv.loc_i = g->loc_i.as<int32_t>();
    tv.st = wp.st; tv.cap = wp.cap; tv.reward = g->reward_dev.as<double>(); tv.act = wp.act_ep;

    std::vector<int8_t> shared_actions((size_t)std::max(1, g->steps));
    {
        AgentStream ag;   // the same stream the action table was built from (DDQNAgent.scala:14,26-32)
        for (int t = 0; t < g->steps; ++t) shared_actions[(size_t)t] = (int8_t)ag.act(hp);
    }
    std::vector<int32_t> batch((size_t)n_agents, hp.batch_size), accum((size_t)n_agents), sizes((size_t)n_agents);
    for (long long k = 0; k < n_agents; ++k) accum[(size_t)k] = g->intensity[(size_t)(k % n)];
    std::vector<int64_t> act_calls((size_t)n_agents);
    std::vector<double> losses((size_t)n_agents);
    long long g_step = 0, last_learn = 0;   // every node of every episode is on the same schedule (one HyperParams record)
    const int rows_per_launch = std::max(1, 65535 / npad);   // grid.y of the network kernel

    for (int t = 0; t < g->steps; ++t) {
        // Transition.s (WorldToy.scala:99): the states as they are before this step
        HIP_TRY(hipMemcpyAsync(g->prev_st.p, wp.st, cells * 5 * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(g->prev_cap.p, wp.cap, cells * sizeof(double), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(g->prev_tcur.p, wp.tcur, cells * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        const int a = shared_actions[(size_t)t];
        if (a == kGreedyAction) {
            // DDQNAgent.actGreedy on each node's own online network: the cells of `rows_per_launch` episodes are handed
            // to the network kernel as so many "countries" of one instance, each with its own network
            for (int e0 = 0; e0 < count; e0 += rows_per_launch) {
                const int m = std::min(rows_per_launch, count - e0);
                const size_t off = (size_t)e0 * npad;
                MlpArgs ma{};
                ma.rows = m * npad; ma.n = m * npad; ma.npad = m * npad; ma.count = 1;
                ma.state_t_max = g->base.t_max;
                ma.st = wp.st + off; ma.cap = wp.cap + off; ma.tcur = wp.tcur + off; ma.plane = cells;
                ma.nl = view.nl; ma.nets = view.nets; ma.net_stride = view.net_stride; ma.per_country = 1;
                ma.net_index = g->net_index.as<int32_t>() + off;
                ma.clip_ptr = view.clip_ptr; ma.clip_stride = view.clip_stride; ma.clip_q = 0.0;
                ma.act_episode = wp.act_ep + off;
                HIP_TRY(launch_mlp(ma, s));
            }
        } else {
            HIP_TRY(hipMemsetAsync(wp.act_ep, a, cells, s));   // one byte per cell: the action everybody drew
        }
        ea.t_begin = t; ea.t_end = t + 1;
        int32_t rc = timed_episode(g, ea, s);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(s));
        rc = agents_observe_instances(agents.p, tv);   // agent.observe(Transition(s, a, r, s2, done)), WorldToy.scala:99
        if (rc) return rc;
        g_step += 1;
        // WorldToy.scala:150-167
        if (g_step - last_learn >= hp.learn_every) {
            rc = epi_agents_replay_size(agents.p, sizes.data());
            if (rc) return rc;
            if ((long long)sizes[0] >= 4LL * hp.batch_size) {
                std::fill(act_calls.begin(), act_calls.end(), (int64_t)(t + 1));   // DDQNAgent.steps: one act() per world step
                rc = epi_agents_learn(agents.p, nullptr, (int32_t)n_agents, batch.data(), accum.data(), act_calls.data(), losses.data(), nullptr);
                if (rc) return rc;
                last_learn = g_step;
            }
        }
    }
    return EPI_OK;
}

// Enqueues init -> episode segments (split around the argmax-Q steps) -> loss for the given grid points on `s`,
// with the per-episode state in `wp`.  Allocates nothing unless `may_allocate` and timing is on (HIP events per launch).
static int32_t grid_enqueue(epi_grid* g, int32_t g_begin, int32_t g_stride, int32_t g_count, const WorkPtrs& wp, double* d_loss,
                            int32_t* d_arr, double* d_pk, hipStream_t s, bool may_allocate) {
    const int n = g->n, npad = g->npad;
    int32_t rc;
    // one outstanding run per handle: the work buffers and the error table are shared, so an earlier run that was
    // only enqueued is waited for and its status collected before they are overwritten (its failure is returned here).
    // The allocation-free entry point may not wait (its callers include code under stream capture, where a
    // device-wide wait is illegal): it refuses instead.
    if (g->pending) {
        if (!may_allocate)
            return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_run_noalloc: this handle still holds an uncollected run: call epi_grid_status() first "
                                              "(this entry point never waits on the device itself)");
        rc = epi_grid_status(g);
        if (rc) return rc;
    }
    InitArgs ia{};
    ia.n = n; ia.npad = npad; ia.count = g_count; ia.seed_idx = g->seed_idx;
    ia.pop = g->graph.pop.as<int32_t>();
    ia.base_cap = g->base.hosp_cap;
    ia.r0 = (int32_t)g->base.r; ia.d0 = (int32_t)g->base.d; ia.v0 = (int32_t)g->base.v; ia.t0 = g->base.t;
    ia.st = wp.st; ia.cap = wp.cap; ia.tcur = wp.tcur;
    ia.arrival = wp.arrival; ia.peak = wp.peak; ia.err = wp.err;
    HIP_TRY(launch_init(ia, s));

    EpisodeArgs ea{};
    ea.n = n; ea.npad = npad; ea.count = g_count;
    ea.seed_idx = g->seed_idx; ea.n_travel = g->n_travel; ea.g_begin = g_begin; ea.g_stride = g_stride;
    ea.step_t_max = g->steps;  // world.stepOne(cfg.stepsPerEpoch), Calibrate.scala:70
    ea.validate = 1;
    ea.uniform_clock = 1;      // every country starts from the one base state (Calibrate.scala:34-45)
    ea.shared_actions = 1;     // the action table of a sweep holds one action per step (grid_build)
    g->graph.fill(ea);
    ea.sumw_seed = g->sumw_seed.as<double>();
    if (g->graph.sparse) {
        ea.qval_seed = g->qseed.as<double>(); ea.seed_deg = g->seed_deg; ea.seed_long = g->seed_long;
        ea.seed_block = g->seed_block;
        ea.exports = wp.exports;
    } else {
        ea.qseed = g->qseed.as<double>(); ea.qtop_seed = g->qtop_seed.as<double>();
    }
    ea.beta0 = g->beta0.as<double>(); ea.beta_rows = g->n_beta > 0 ? g->n_beta : 1;
    ea.cst = g->cst.as<double>();
    ea.actions = g->actions.as<int8_t>(); ea.act_t0 = 0;
    ea.noise = g->noise.as<double>(); ea.noise_t0 = 0;
    ea.st = ia.st; ea.cap = ia.cap; ea.tcur = ia.tcur; ea.arrival = ia.arrival; ea.peak = ia.peak; ea.err = ia.err;
    if (g->stats.p) ea.stats = g->stats.as<unsigned long long>();
    auto episode = [&](const EpisodeArgs& e) -> int32_t {
        if (may_allocate) return timed_episode(g, e, s);
        HIP_TRY(launch_episode(e, s));
        return EPI_OK;
    };

    if (g->learning) {
        if (!may_allocate)
            return fail(EPI_ERR_UNSUPPORTED, "epi_grid_run_noalloc: this sweep's rollouts learn (one agent per grid point and country is "
                                             "allocated per call): use epi_grid_run_strided");
        rc = grid_run_learning(g, ea, wp, s);
        if (rc) return rc;
    }
    int t0 = g->learning ? g->steps : 0;
    for (int gs : g->greedy_steps) {
        if (g->learning) break;
        if (gs > t0) {
            ea.t_begin = t0; ea.t_end = gs; ea.act_episode = nullptr;
            rc = episode(ea);
            if (rc) return rc;
        }
        MlpArgs ma{};
        ma.rows = g_count * n; ma.n = n; ma.npad = npad; ma.count = g_count;
        ma.state_t_max = g->base.t_max;
        ma.st = ia.st; ma.cap = ia.cap; ma.tcur = ia.tcur; ma.plane = (size_t)g_count * npad;
        g->mlp.fill(ma);
        ma.act_episode = wp.act_ep;
        HIP_TRY(launch_mlp(ma, s));
        ea.t_begin = gs; ea.t_end = gs + 1; ea.act_episode = ma.act_episode;
        rc = episode(ea);
        if (rc) return rc;
        t0 = gs + 1;
    }
    if (t0 < g->steps) {
        ea.t_begin = t0; ea.t_end = g->steps; ea.act_episode = nullptr;
        rc = episode(ea);
        if (rc) return rc;
    }

    LossArgs la{};
    la.n = n; la.npad = npad; la.count = g_count; la.steps = g->steps;
    la.arrival = ia.arrival; la.peak = ia.peak;
    la.t_arrive = g->t_arrive.as<double>(); la.has_arrive = g->has_arrive.as<uint8_t>();
    la.t_peak = g->t_peak.as<double>(); la.has_peak = g->has_peak.as<uint8_t>();
    la.loss = d_loss; la.arrival_out = d_arr; la.peak_out = d_pk;
    la.id_of = g->graph.renumbered ? g->graph.d_id_of.as<int32_t>() : nullptr;
    HIP_TRY(launch_loss(la, s));

    g->pending = true;
    g->last_count = g_count;
    g->last_g_begin = g_begin;
    g->last_g_stride = g_stride;
    g->last_err = wp.err;
    return EPI_OK;
}

static int32_t grid_check_points(const epi_grid* g, int32_t g_begin, int32_t g_stride, int32_t g_count, const char* who) {
    const int G = g->n_beta * g->n_travel;
    if (g_begin < 0 || g_count < 0 || g_stride < 1 || (g_count > 0 && (int64_t)g_begin + (int64_t)(g_count - 1) * g_stride >= G))
        return fail(EPI_ERR_BAD_ARGUMENT, "%s: points %d + k*%d, k < %d, leave the %d grid points", who, g_begin, g_stride, g_count, G);
    return EPI_OK;
}

EPI_API int32_t epi_grid_run_strided(epi_grid* g, int32_t g_begin, int32_t g_stride, int32_t g_count, double* loss,
                                     int32_t* arrival, double* peak, int32_t out_on_device, void* stream) {
    if (!g) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_run: null handle");
    int32_t rc = grid_check_points(g, g_begin, g_stride, g_count, "epi_grid_run");
    if (rc) return rc;
    if (g_count == 0) return EPI_OK;
    HIP_TRY(hipSetDevice(g->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = g->work.ensure(g_count, g->npad, g->graph.sparse);
    if (rc) return rc;
    const int n = g->n;

    double* d_loss = loss;
    int32_t* d_arr = arrival;
    double* d_pk = peak;
    if (!out_on_device) {
        HIP_TRY(g->loss_tmp.ensure(sizeof(double) * (size_t)g_count));
        d_loss = g->loss_tmp.as<double>();
        if (arrival) { HIP_TRY(g->arrival_tmp.ensure(sizeof(int32_t) * (size_t)g_count * n)); d_arr = g->arrival_tmp.as<int32_t>(); }
        if (peak) { HIP_TRY(g->peak_tmp.ensure(sizeof(double) * (size_t)g_count * n)); d_pk = g->peak_tmp.as<double>(); }
    }
    rc = grid_enqueue(g, g_begin, g_stride, g_count, work_ptrs(g->work), d_loss, d_arr, d_pk, s, true);
    if (rc) return rc;
    if (out_on_device) return EPI_OK;

    HIP_TRY(hipStreamSynchronize(s));
    if (loss) HIP_TRY(hipMemcpy(loss, d_loss, sizeof(double) * (size_t)g_count, hipMemcpyDeviceToHost));
    if (arrival) HIP_TRY(hipMemcpy(arrival, d_arr, sizeof(int32_t) * (size_t)g_count * n, hipMemcpyDeviceToHost));
    if (peak) HIP_TRY(hipMemcpy(peak, d_pk, sizeof(double) * (size_t)g_count * n, hipMemcpyDeviceToHost));
    return epi_grid_status(g);
}

// The sweep of epi_grid_run_strided(out_on_device = 1) without any allocation: the per-episode state (compartments,
// hospCap, clocks, arrival / peak, error table, argmax-Q actions, sparse export lists) lives in the fixed work area
// g_noalloc_scratch instead of buffers the handle allocates on demand.  Every call on a GPU shares that GPU's array
// and overwrites it from the start (the init kernel writes every word the episode kernels read), so two calls that
// overlap in time on one GPU -- different streams, threads or handles -- race; calls on one stream are ordered.
EPI_API int32_t epi_grid_run_noalloc(epi_grid* g, int32_t g_begin, int32_t g_stride, int32_t g_count, double* loss_dev,
                                     int32_t* arrival_dev, double* peak_dev, void* stream) {
    if (!g) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_run_noalloc: null handle");
    int32_t rc = grid_check_points(g, g_begin, g_stride, g_count, "epi_grid_run_noalloc");
    if (rc) return rc;
    if (g_count == 0) return EPI_OK;
    if (!loss_dev) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_run_noalloc: loss_dev must be device memory for %d doubles", g_count);
    const size_t need = scratch_need((size_t)g_count, (size_t)g->npad, g->graph.sparse);
    if (need > kNoallocScratchBytes)
        return fail(EPI_ERR_UNSUPPORTED,
                    "epi_grid_run_noalloc: %d episodes of %d padded countries need %zu bytes of work area, the fixed one holds %zu "
                    "(at most %zu episodes of this world per call): split the call or use epi_grid_run_strided",
                    g_count, g->npad, need, kNoallocScratchBytes,
                    (size_t)(kNoallocScratchBytes / (scratch_need(1, (size_t)g->npad, g->graph.sparse) + 1)));
    unsigned char* base = nullptr;
    rc = scratch_base(g->device, &base);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(g->device));
    WorkPtrs wp = carve_scratch(base, (size_t)g_count, (size_t)g->npad, g->graph.sparse);
    if (g_count > g->noalloc_max_episodes)   // cannot happen after the size check above; keeps the table's bound explicit
        return fail(EPI_ERR_UNSUPPORTED, "epi_grid_run_noalloc: %d episodes, the handle's error table holds %d", g_count, g->noalloc_max_episodes);
    wp.err = g->noalloc_err.as<int32_t>();   // the handle's own table, not the shared area's
    return grid_enqueue(g, g_begin, g_stride, g_count, wp, loss_dev, arrival_dev, peak_dev, static_cast<hipStream_t>(stream), false);
}

EPI_API int32_t epi_calibrate_grid(const epi_grid_desc* desc, double* loss, int32_t* arrival, double* peak, int32_t* best_idx,
                                   double* best_loss) {
    if (!desc) return fail(EPI_ERR_BAD_ARGUMENT, "epi_calibrate_grid: null descriptor");
    const int G = desc->n_beta * desc->n_travel;
    std::vector<double> tmp;
    if (!loss) { tmp.resize((size_t)(G > 0 ? G : 1)); loss = tmp.data(); }
    epi_grid* g = epi_grid_create(desc);
    if (!g) return g_last_code ? g_last_code : EPI_ERR_DEVICE;
    int32_t rc = epi_grid_run(g, 0, G, loss, arrival, peak, 0, nullptr);
    epi_grid_destroy(g);
    if (rc) return rc;
    // bestPair defaults to the first grid point (Calibrate.scala:97): index 0
    return epi_grid_argmin(loss, G, best_idx, best_loss);
}

// ================================================================================================
// graph construction
// ================================================================================================
EPI_API int32_t epi_gravity_matrix(int32_t n, const int32_t* pop, const double* lat, const double* lon, const double* flight_mul,
                                   double a, double b, double g, double k, double cap_row, double* W, int32_t device) {
    if (n < 0 || (n > 0 && (!pop || !lat || !lon || !W))) return fail(EPI_ERR_BAD_ARGUMENT, "epi_gravity_matrix: null argument");
    if (n == 0) return EPI_OK;
    for (int i = 0; i < n; ++i)
        if (pop[i] < 0 || !std::isfinite(lat[i]) || !std::isfinite(lon[i]))
            return fail(EPI_ERR_BAD_ARGUMENT, "epi_gravity_matrix: country %d: pop must be >= 0 and lat/lon finite", i);
    int32_t rc = use_device(device);
    if (rc) return rc;
    const size_t nn = (size_t)n * n;
    DevBuf dpop, dlat, dlon, dfl, dW;
    HIP_TRY(dpop.upload(std::vector<int32_t>(pop, pop + n)));
    HIP_TRY(dlat.upload(std::vector<double>(lat, lat + n)));
    HIP_TRY(dlon.upload(std::vector<double>(lon, lon + n)));
    if (flight_mul) {
        HIP_TRY(dfl.ensure(nn * sizeof(double)));
        HIP_TRY(hipMemcpy(dfl.p, flight_mul, nn * sizeof(double), hipMemcpyHostToDevice));
    }
    HIP_TRY(dW.ensure(nn * sizeof(double)));
    GravityArgs ga{};
    ga.n = n; ga.pop = dpop.as<int32_t>(); ga.lat = dlat.as<double>(); ga.lon = dlon.as<double>();
    ga.flight = flight_mul ? dfl.as<double>() : nullptr;
    ga.a = a; ga.b = b; ga.g = g; ga.k = k; ga.cap_row = cap_row;
    ga.W = dW.as<double>();
    HIP_TRY(launch_gravity(ga, nullptr));
    HIP_TRY(hipMemcpy(W, dW.p, nn * sizeof(double), hipMemcpyDeviceToHost));
    return EPI_OK;
}

EPI_API int32_t epi_gravity_csr(int32_t n, const int32_t* pop, const double* lat, const double* lon, const double* flight_mul,
                                double a, double b, double g, double k, double cap_row, double threshold, int32_t* rowptr,
                                int32_t* col, double* val, int32_t device) {
    if (n < 0 || (n > 0 && (!pop || !lat || !lon)) || !rowptr) return fail(EPI_ERR_BAD_ARGUMENT, "epi_gravity_csr: null argument");
    if ((col == nullptr) != (val == nullptr)) return fail(EPI_ERR_BAD_ARGUMENT, "epi_gravity_csr: col and val come together");
    rowptr[0] = 0;
    if (n == 0) return EPI_OK;
    for (int i = 0; i < n; ++i)
        if (pop[i] < 0 || !std::isfinite(lat[i]) || !std::isfinite(lon[i]))
            return fail(EPI_ERR_BAD_ARGUMENT, "epi_gravity_csr: country %d: pop must be >= 0 and lat/lon finite", i);
    if (!(threshold >= 0.0)) return fail(EPI_ERR_BAD_ARGUMENT, "epi_gravity_csr: threshold %g must be >= 0", threshold);
    int32_t rc = use_device(device);
    if (rc) return rc;
    DevBuf dpop, dlat, dlon, dfl, dscale, dcnt, drp, dcol, dval;
    HIP_TRY(dpop.upload(std::vector<int32_t>(pop, pop + n)));
    HIP_TRY(dlat.upload(std::vector<double>(lat, lat + n)));
    HIP_TRY(dlon.upload(std::vector<double>(lon, lon + n)));
    if (flight_mul) {
        HIP_TRY(dfl.ensure((size_t)n * n * sizeof(double)));
        HIP_TRY(hipMemcpy(dfl.p, flight_mul, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice));
    }
    HIP_TRY(dscale.ensure((size_t)n * sizeof(double)));
    HIP_TRY(dcnt.ensure((size_t)n * sizeof(int32_t)));
    GravityCsrArgs ca{};
    ca.g.n = n; ca.g.pop = dpop.as<int32_t>(); ca.g.lat = dlat.as<double>(); ca.g.lon = dlon.as<double>();
    ca.g.flight = flight_mul ? dfl.as<double>() : nullptr;
    ca.g.a = a; ca.g.b = b; ca.g.g = g; ca.g.k = k; ca.g.cap_row = cap_row; ca.g.W = nullptr;
    ca.threshold = threshold;
    ca.scale = dscale.as<double>(); ca.counts = dcnt.as<int32_t>();
    ca.mode = 0;
    HIP_TRY(launch_gravity_csr(ca, nullptr));
    ca.mode = 1;
    HIP_TRY(launch_gravity_csr(ca, nullptr));
    std::vector<int32_t> cnt((size_t)n);
    HIP_TRY(hipMemcpy(cnt.data(), dcnt.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        total += cnt[i];
        if (total > 2147483647LL) return fail(EPI_ERR_UNSUPPORTED, "epi_gravity_csr: more than 2^31 - 1 stored entries");
        rowptr[i + 1] = (int32_t)total;
    }
    if (!col || total == 0) return EPI_OK;   // first call: the caller sizes col / val from rowptr[n]
    HIP_TRY(drp.upload(std::vector<int32_t>(rowptr, rowptr + n + 1)));
    HIP_TRY(dcol.ensure((size_t)total * sizeof(int32_t)));
    HIP_TRY(dval.ensure((size_t)total * sizeof(double)));
    ca.mode = 2; ca.rowptr = drp.as<int32_t>(); ca.col = dcol.as<int32_t>(); ca.val = dval.as<double>();
    HIP_TRY(launch_gravity_csr(ca, nullptr));
    HIP_TRY(hipMemcpy(col, dcol.p, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(val, dval.p, (size_t)total * sizeof(double), hipMemcpyDeviceToHost));
    return EPI_OK;
}

// ================================================================================================
// batched worlds
// ================================================================================================
static int32_t world_build(epi_world* Wd, const epi_world_desc* d) {
    if (d->n < 1 || d->batch < 1) return fail(EPI_ERR_BAD_ARGUMENT, "world: n = %d, batch = %d", d->n, d->batch);
    if (!d->pop || !d->env || !d->prof || d->seed_idx < 0 || d->seed_idx >= d->n) return fail(EPI_ERR_BAD_ARGUMENT, "world: null input or bad seed index");
    int32_t rc = check_base_state(d->base, "world");
    if (rc) return rc;
    if (d->validate) {
        rc = assert_initial(d->base);
        if (rc) return rc;
    }
    rc = use_device(d->device);
    if (rc) return rc;
    HIP_TRY(prepare_device_kernels());
    Wd->device = d->device;
    Wd->n = d->n; Wd->batch = d->batch; Wd->seed_idx = d->seed_idx; Wd->validate = d->validate;
    Wd->base = d->base;
    rc = Wd->graph.build(d->n, d->pop, d->conn, d->csr_rowptr, d->csr_col, d->csr_val);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    Wd->graph.d_W.release();   // the tables are complete: the matrix itself is not read again
    Wd->graph.W.clear(); Wd->graph.W.shrink_to_fit();
    const int n = d->n, npad = Wd->graph.npad;
    Wd->npad = npad;
    std::vector<double> hb((size_t)npad, 0.0), hc((size_t)CST_COUNT * npad, 0.0);
    std::vector<RewardConsts> hr((size_t)npad);
    std::memset(hr.data(), 0, hr.size() * sizeof(RewardConsts));
    for (int c = 0; c < n; ++c) {
        const StepConsts sc = make_step_consts(d->env[c], d->prof[c]);
        hb[c] = sc.beta0;
        hc[(size_t)CST_GAMMA * npad + c] = sc.gamma;
        hc[(size_t)CST_IFR0 * npad + c] = sc.ifr0;
        hc[(size_t)CST_CNOISE * npad + c] = sc.cnoise;
        hc[(size_t)CST_EFF1 * npad + c] = sc.eff1;
        hc[(size_t)CST_EFF2 * npad + c] = sc.eff2;
        hc[(size_t)CST_EFF3 * npad + c] = sc.eff3;
        hc[(size_t)CST_SURGE * npad + c] = sc.surge;
        hc[(size_t)CST_VAX6 * npad + c] = sc.vax6;
        hc[(size_t)CST_VAX7 * npad + c] = sc.vax7;
        hr[c] = make_reward_consts(d->env[c], d->prof[c]);
    }
    HIP_TRY(Wd->beta0.upload(hb));
    HIP_TRY(Wd->cst.upload(hc));
    HIP_TRY(Wd->rwd.upload(hr));
    Wd->env_seeds.assign((size_t)n, 42);  // EpidemicEnv default seed (env.scala:40)
    if (d->env_seeds) Wd->env_seeds.assign(d->env_seeds, d->env_seeds + n);
    Wd->env_rnd.clear();
    for (int c = 0; c < n; ++c) Wd->env_rnd.emplace_back((int64_t)Wd->env_seeds[c]);
    rc = Wd->work.ensure(d->batch, npad, Wd->graph.sparse);
    if (rc) return rc;
    const size_t cells = (size_t)d->batch * npad;
    HIP_TRY(Wd->act_dev.ensure(cells));
    HIP_TRY(Wd->gauss_dev.ensure(cells * sizeof(double)));
    HIP_TRY(Wd->reward_dev.ensure(cells * sizeof(double)));
    HIP_TRY(Wd->flow_s.ensure(cells * sizeof(int32_t)));
    HIP_TRY(Wd->flow_i.ensure(cells * sizeof(int32_t)));
    if (cells > (size_t)1 << 30) return fail(EPI_ERR_UNSUPPORTED, "world: batch x padded countries = %zu exceeds 2^30 cells", cells);
    const size_t compact = (size_t)d->batch * n;
    HIP_TRY(Wd->pin_act.ensure(compact * sizeof(int32_t)));
    HIP_TRY(Wd->pin_gauss.ensure(compact * sizeof(double)));
    HIP_TRY(Wd->pin_states.ensure(compact * sizeof(epi_state)));
    HIP_TRY(Wd->pin_packed.ensure(compact * sizeof(epi_state_packed)));
    HIP_TRY(Wd->pin_rewards.ensure(compact * sizeof(double)));
    HIP_TRY(Wd->pin_actions.ensure(compact * sizeof(int32_t)));
    HIP_TRY(Wd->pin_err.ensure((size_t)d->batch * 4 * sizeof(int32_t)));
    return EPI_OK;
}

static int32_t world_init_states(epi_world* w) {
    InitArgs ia{};
    ia.n = w->n; ia.npad = w->npad; ia.count = w->batch; ia.seed_idx = w->seed_idx;
    ia.pop = w->graph.pop.as<int32_t>();
    ia.base_cap = w->base.hosp_cap;
    ia.r0 = (int32_t)w->base.r; ia.d0 = (int32_t)w->base.d; ia.v0 = (int32_t)w->base.v; ia.t0 = w->base.t;
    ia.st = w->work.st.as<int32_t>(); ia.cap = w->work.cap.as<double>(); ia.tcur = w->work.tcur.as<int32_t>();
    ia.arrival = w->work.arrival.as<int32_t>(); ia.peak = w->work.peak.as<int32_t>(); ia.err = w->work.err.as<int32_t>();
    HIP_TRY(launch_init(ia, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    return EPI_OK;
}

EPI_API epi_world* epi_world_create(const epi_world_desc* desc) {
    if (!desc) { fail(EPI_ERR_BAD_ARGUMENT, "epi_world_create: null descriptor"); return nullptr; }
    std::unique_ptr<epi_world> w(new epi_world());
    if (world_build(w.get(), desc) != EPI_OK) return nullptr;
    if (world_init_states(w.get()) != EPI_OK) return nullptr;
    return w.release();
}

EPI_API int32_t epi_world_destroy(epi_world* w) {
    if (!w) return EPI_OK;
    (void)hipSetDevice(w->device);
    delete w;
    return EPI_OK;
}

EPI_API int32_t epi_world_reset(epi_world* w) {
    if (!w) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_reset: null handle");
    HIP_TRY(hipSetDevice(w->device));
    w->have_tr = false;   // the kept transition belongs to states that no longer exist
    return world_init_states(w);
}

static void world_io_args(const epi_world* w, WorldIoArgs& io) {
    io.n = w->n; io.npad = w->npad; io.batch = w->batch;
    io.act_pad = w->act_dev.as<int8_t>(); io.gauss_pad = w->gauss_dev.as<double>();
    io.err = w->work.err.as<int32_t>();
    io.st = w->work.st.as<int32_t>(); io.cap = w->work.cap.as<double>(); io.tcur = w->work.tcur.as<int32_t>();
    io.reward_pad = w->reward_dev.as<double>();
    io.base = w->base;
}

static int32_t world_download_states(epi_world* w, epi_state* out) {
    WorldIoArgs io{};
    world_io_args(w, io);
    io.states_out = w->pin_states.on_device<epi_state>();
    const size_t bytes = (size_t)w->batch * w->n * sizeof(epi_state);
    HIP_TRY(launch_world_unpack(io, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    std::memcpy(out, w->pin_states.p, bytes);
    return EPI_OK;
}

EPI_API int32_t epi_world_result_buffers(epi_world* w, epi_state** states, double** rewards, int32_t** actions) {
    if (!w) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_result_buffers: null handle");
    if (states) *states = w->pin_states.as<epi_state>();
    if (rewards) *rewards = w->pin_rewards.as<double>();
    if (actions) *actions = w->pin_actions.as<int32_t>();
    return EPI_OK;
}

EPI_API int32_t epi_world_get_states(epi_world* w, epi_state* out) {
    if (!w || !out) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_get_states: null argument");
    HIP_TRY(hipSetDevice(w->device));
    return world_download_states(w, out);
}

EPI_API int32_t epi_world_set_states(epi_world* w, const epi_state* in) {
    if (!w || !in) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_set_states: null argument");
    HIP_TRY(hipSetDevice(w->device));
    const int n = w->n, npad = w->npad, B = w->batch;
    const size_t cells = (size_t)B * npad;
    std::vector<int32_t> st(cells * 5, 0), tc(cells, 0);
    std::vector<double> cap(cells, 1.0);
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < n; ++c) {
            const epi_state& s = in[(size_t)b * n + c];
            if (!is_count(s.s) || !is_count(s.i) || !is_count(s.r) || !is_count(s.d) || !is_count(s.v) || !std::isfinite(s.hosp_cap))
                return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_set_states: instance %d country %d: compartments must be non-negative integers below 2^31", b, c);
            const size_t idx = (size_t)b * npad + c;
            st[0 * cells + idx] = (int32_t)s.s; st[1 * cells + idx] = (int32_t)s.i; st[2 * cells + idx] = (int32_t)s.r;
            st[3 * cells + idx] = (int32_t)s.d; st[4 * cells + idx] = (int32_t)s.v;
            cap[idx] = s.hosp_cap;
            tc[idx] = s.t;
        }
    HIP_TRY(hipMemcpy(w->work.st.p, st.data(), st.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(w->work.tcur.p, tc.data(), tc.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(w->work.cap.p, cap.data(), cap.size() * sizeof(double), hipMemcpyHostToDevice));
    w->have_tr = false;   // the kept transition's s2 planes were just overwritten
    return EPI_OK;
}

EPI_API int32_t epi_world_set_mlp(epi_world* w, int32_t hidden, const double* W1, const double* b1, const double* W2,
                                  const double* b2, const double* W3, const double* b3, double clip_q) {
    if (!w || !W1 || !W2 || !W3) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_set_mlp: null argument");
    HIP_TRY(hipSetDevice(w->device));
    int32_t rc = w->mlp.upload(hidden, W1, b1, W2, b2, W3, b3, clip_q);
    if (rc) return rc;
    w->mlp_set = true;
    return EPI_OK;
}

static int32_t world_step_impl(epi_world* w, const int32_t* actions, bool greedy, const double* gauss, int32_t t_max,
                               epi_state* out_states, double* out_rewards, int32_t* out_actions,
                               epi_state_packed* out_packed = nullptr) {
    const int n = w->n, npad = w->npad, B = w->batch;
    const size_t cells = (size_t)B * npad, compact = (size_t)B * n;
    HIP_TRY(hipSetDevice(w->device));
    if (greedy && !w->mlp_set) return fail(EPI_ERR_BAD_ARGUMENT, "world greedy step: call epi_world_set_mlp first");
    hipStream_t s = nullptr;
    WorldIoArgs io{};
    world_io_args(w, io);
    // Inputs are placed in mapped page-locked memory and read by the pack kernel over PCIe: no copy calls.
    // noise draws: caller-supplied, or one JVM-compatible stream per country shared by all instances
    double* hz = w->pin_gauss.as<double>();
    if (gauss) {
        for (size_t k = 0; k < compact; ++k) {
            if (!std::isfinite(gauss[k]))
                return fail(EPI_ERR_BAD_ARGUMENT, "world step: gauss[%d][%d] is not finite", (int)(k / n), (int)(k % n));
            hz[k] = gauss[k];
        }
        io.gauss = w->pin_gauss.on_device<double>();
    } else {
        for (int c = 0; c < n; ++c) hz[c] = w->env_rnd[c].nextGaussian();
        io.gauss_row = w->pin_gauss.on_device<double>();
    }
    if (!greedy) {
        std::memcpy(w->pin_act.p, actions, compact * sizeof(int32_t));
        io.actions = w->pin_act.on_device<int32_t>();
    }
    // Small worlds (the toy world of the batched DDQN rollout): the episode kernel reads the compact inputs and writes
    // the compact results itself -- one launch per step instead of pack + episode + unpack.
    const bool fused = !greedy && !w->graph.sparse && n <= 16 && npad == 64 && !w->keep_tr;
    if (w->keep_tr) {   // the Transition's `s` (WorldToy.scala:99): the states as they are before this step
        HIP_TRY(hipMemcpyAsync(w->prev_st.p, w->work.st.p, cells * 5 * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(w->prev_cap.p, w->work.cap.p, cells * sizeof(double), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(w->prev_tcur.p, w->work.tcur.p, cells * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    }
    if (!fused) HIP_TRY(launch_world_pack(io, s));
    if (greedy) {
        MlpArgs ma{};
        ma.rows = B * n; ma.n = n; ma.npad = npad; ma.count = B;
        ma.state_t_max = w->base.t_max;
        ma.st = w->work.st.as<int32_t>(); ma.cap = w->work.cap.as<double>(); ma.tcur = w->work.tcur.as<int32_t>();
        ma.plane = cells;
        w->mlp.fill(ma);
        ma.act_episode = w->act_dev.as<int8_t>();
        HIP_TRY(launch_mlp(ma, s));
    }
    EpisodeArgs ea{};
    ea.n = n; ea.npad = npad; ea.count = B;
    ea.seed_idx = w->seed_idx; ea.n_travel = 0; ea.g_begin = 0; ea.g_stride = 1;
    ea.t_begin = 0; ea.t_end = 1;
    ea.step_t_max = t_max;
    ea.validate = w->validate;
    w->graph.fill(ea);
    ea.qseed = nullptr; ea.sumw_seed = nullptr; ea.qtop_seed = nullptr; ea.qval_seed = nullptr;
    if (w->graph.sparse) ea.exports = w->work.exports.as<int32_t>();
    ea.beta0 = w->beta0.as<double>(); ea.beta_rows = 1;
    ea.cst = w->cst.as<double>();
    ea.actions = nullptr; ea.act_episode = w->act_dev.as<int8_t>();
    ea.noise = nullptr; ea.noise_episode = w->gauss_dev.as<double>();
    ea.st = w->work.st.as<int32_t>(); ea.cap = w->work.cap.as<double>(); ea.tcur = w->work.tcur.as<int32_t>();
    ea.arrival = w->work.arrival.as<int32_t>(); ea.peak = w->work.peak.as<int32_t>(); ea.err = w->work.err.as<int32_t>();
    ea.rwd = w->rwd.as<RewardConsts>(); ea.reward_out = w->reward_dev.as<double>();
    ea.flow_s = w->flow_s.as<int32_t>(); ea.flow_i = w->flow_i.as<int32_t>();
    if (w->keep_tr) { ea.loc_s = w->loc_s.as<int32_t>(); ea.loc_i = w->loc_i.as<int32_t>(); }
    if (fused) {
        ea.io_actions = io.actions; ea.io_gauss = io.gauss; ea.io_gauss_row = io.gauss_row;
        ea.io_states = out_states ? w->pin_states.on_device<epi_state>() : nullptr;
        ea.io_packed = out_packed ? w->pin_packed.on_device<epi_state_packed>() : nullptr;
        ea.io_rewards = out_rewards ? w->pin_rewards.on_device<double>() : nullptr;
        ea.io_err = w->pin_err.on_device<int32_t>();
        ea.io_t_max = w->base.t_max; ea.io_seed = w->base.seed;
        ea.act_episode = nullptr;
    }
    if (w->timing) HIP_TRY(hipEventRecord(w->ev0, s));
    HIP_TRY(launch_episode(ea, s));
    if (w->timing) HIP_TRY(hipEventRecord(w->ev1, s));
    // Results: the unpack kernel writes compact records straight into the handle's page-locked, device-mapped
    // result arrays; a caller that reads them in place (epi_world_result_buffers) gets no further copy.
    if (out_states) io.states_out = w->pin_states.on_device<epi_state>();
    if (out_packed) io.packed_out = w->pin_packed.on_device<epi_state_packed>();
    if (out_rewards) io.rewards_out = w->pin_rewards.on_device<double>();
    if (out_actions) io.actions_out = w->pin_actions.on_device<int32_t>();
    io.err_out = w->pin_err.on_device<int32_t>();
    if (!fused) HIP_TRY(launch_world_unpack(io, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (w->timing) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, w->ev0, w->ev1));
        w->kernel_ms += ms;
        w->kernel_launches += 1;
        w->kernel_country_steps += (long long)B * n;
    }
    w->have_flows = true;
    w->have_tr = w->keep_tr;
    if (out_states && out_states != w->pin_states.p) std::memcpy(out_states, w->pin_states.p, compact * sizeof(epi_state));
    if (out_packed && out_packed != w->pin_packed.p) std::memcpy(out_packed, w->pin_packed.p, compact * sizeof(epi_state_packed));
    if (out_rewards && out_rewards != w->pin_rewards.p) std::memcpy(out_rewards, w->pin_rewards.p, compact * sizeof(double));
    if (out_actions && out_actions != w->pin_actions.p) std::memcpy(out_actions, w->pin_actions.p, compact * sizeof(int32_t));
    const int32_t* he = w->pin_err.as<int32_t>();
    for (int e = 0; e < B; ++e)
        if (he[(size_t)e * 4] != 0)
            return fail(he[(size_t)e * 4], "%s (instance %d, country %d, t=%d)", status_text(he[(size_t)e * 4]), e,
                        he[(size_t)e * 4 + 1], he[(size_t)e * 4 + 2]);
    return EPI_OK;
}

EPI_API int32_t epi_world_step(epi_world* w, const int32_t* actions, const double* gauss, int32_t t_max,
                               epi_state* out_states, double* out_rewards) {
    if (!w || !actions) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_step: null argument");
    return world_step_impl(w, actions, false, gauss, t_max, out_states, out_rewards, nullptr);
}

EPI_API int32_t epi_world_step_packed(epi_world* w, const int32_t* actions, const double* gauss, int32_t t_max,
                                      epi_state_packed* out_states, double* out_rewards) {
    if (!w || !actions) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_step_packed: null argument");
    return world_step_impl(w, actions, false, gauss, t_max, nullptr, out_rewards, nullptr, out_states);
}

EPI_API int32_t epi_world_packed_buffer(epi_world* w, epi_state_packed** states) {
    if (!w || !states) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_packed_buffer: null argument");
    *states = w->pin_packed.as<epi_state_packed>();
    return EPI_OK;
}

EPI_API int32_t epi_world_step_greedy(epi_world* w, const double* gauss, int32_t t_max, epi_state* out_states,
                                      double* out_rewards, int32_t* out_actions) {
    if (!w) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_step_greedy: null handle");
    return world_step_impl(w, nullptr, true, gauss, t_max, out_states, out_rewards, out_actions);
}

// `steps` consecutive WorldToy.stepOneGreedy calls without leaving the device (the trainer's evaluation loop,
// MutliTrainer.scala:137-141): per step one launch of the network kernel (DDQNAgent.actGreedy for every instance and
// country) and one of the episode kernel, all enqueued back to back; one synchronisation at the end.
EPI_API int32_t epi_world_rollout_greedy(epi_world* w, epi_agents* agents, int32_t steps, int32_t t_max, double* i_series,
                                         int32_t* arrival, double* peak, epi_state* out_states) {
    if (!w) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_rollout_greedy: null handle");
    if (steps < 0) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_rollout_greedy: steps = %d", steps);
    const int n = w->n, npad = w->npad, B = w->batch;
    const size_t cells = (size_t)B * npad, compact = (size_t)B * n;
    AgentsNetView view{};
    if (agents) {
        int32_t rc = agents_net_view(agents, &view);
        if (rc) return rc;
        if (view.device != w->device) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_rollout_greedy: the agents live on device %d, the world on device %d", view.device, w->device);
        if (view.n_agents < n) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_rollout_greedy: %d agents for %d countries (one agent per node, WorldToy.scala:50)", view.n_agents, n);
    } else if (!w->mlp_set) {
        return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_rollout_greedy: pass the nodes' agents or call epi_world_set_mlp first");
    }
    HIP_TRY(hipSetDevice(w->device));
    w->have_tr = false;   // stepOneGreedy observes nothing (WorldToy.scala:172-235): no transition is kept past this call
    hipStream_t s = nullptr;
    // Validation.arrivalTime / peakI count from the first step of this call; the error table starts clean
    HIP_TRY(hipMemsetAsync(w->work.arrival.p, 0xff, cells * sizeof(int32_t), s));
    HIP_TRY(hipMemsetAsync(w->work.peak.p, 0, cells * sizeof(int32_t), s));
    HIP_TRY(hipMemsetAsync(w->work.err.p, 0, (size_t)B * 4 * sizeof(int32_t), s));
    if (i_series && steps > 0) HIP_TRY(w->series_dev.ensure(sizeof(double) * (size_t)steps * compact));
    // env noise: the next `steps` draws of every country's own stream (env.scala:42,91), a table [chunk][npad]
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, steps), ((size_t)8 << 20) / (size_t)npad));
    std::vector<double> hz((size_t)chunk * npad, 0.0);
    HIP_TRY(w->noise_tab.ensure(hz.size() * sizeof(double)));

    MlpArgs ma{};
    ma.rows = B * n; ma.n = n; ma.npad = npad; ma.count = B;
    ma.state_t_max = w->base.t_max;
    ma.st = w->work.st.as<int32_t>(); ma.cap = w->work.cap.as<double>(); ma.tcur = w->work.tcur.as<int32_t>();
    ma.plane = cells;
    if (agents) {
        ma.nl = view.nl; ma.nets = view.nets; ma.net_stride = view.net_stride; ma.per_country = 1;
        ma.clip_ptr = view.clip_ptr; ma.clip_stride = view.clip_stride; ma.clip_q = 0.0;
    } else {
        w->mlp.fill(ma);
    }
    ma.act_episode = w->act_dev.as<int8_t>();

    EpisodeArgs ea{};
    ea.n = n; ea.npad = npad; ea.count = B;
    ea.seed_idx = w->seed_idx; ea.n_travel = 0; ea.g_begin = 0; ea.g_stride = 1;
    ea.step_t_max = t_max;
    ea.validate = w->validate;
    w->graph.fill(ea);
    ea.qseed = nullptr; ea.sumw_seed = nullptr; ea.qtop_seed = nullptr; ea.qval_seed = nullptr;
    if (w->graph.sparse) ea.exports = w->work.exports.as<int32_t>();
    ea.beta0 = w->beta0.as<double>(); ea.beta_rows = 1;
    ea.cst = w->cst.as<double>();
    ea.actions = nullptr; ea.act_episode = w->act_dev.as<int8_t>();
    ea.noise = w->noise_tab.as<double>(); ea.noise_episode = nullptr;
    ea.st = w->work.st.as<int32_t>(); ea.cap = w->work.cap.as<double>(); ea.tcur = w->work.tcur.as<int32_t>();
    ea.arrival = w->work.arrival.as<int32_t>(); ea.peak = w->work.peak.as<int32_t>(); ea.err = w->work.err.as<int32_t>();
    ea.rwd = w->rwd.as<RewardConsts>(); ea.reward_out = w->reward_dev.as<double>();
    ea.flow_s = w->flow_s.as<int32_t>(); ea.flow_i = w->flow_i.as<int32_t>();
    ea.series_i = i_series ? w->series_dev.as<double>() : nullptr;
    ea.series_t0 = 0;
    for (int k0 = 0; k0 < steps; k0 += chunk) {
        const int kn = std::min(chunk, steps - k0);
        if (k0 > 0) HIP_TRY(hipStreamSynchronize(s));   // the previous chunk's table is still being read
        for (int k = 0; k < kn; ++k)
            for (int c = 0; c < n; ++c) hz[(size_t)k * npad + c] = w->env_rnd[c].nextGaussian();
        HIP_TRY(hipMemcpyAsync(w->noise_tab.p, hz.data(), (size_t)kn * npad * sizeof(double), hipMemcpyHostToDevice, s));
        ea.noise_t0 = k0;
        for (int k = k0; k < k0 + kn; ++k) {
            HIP_TRY(launch_mlp(ma, s));
            ea.t_begin = k; ea.t_end = k + 1;
            HIP_TRY(launch_episode(ea, s));
        }
        HIP_TRY(hipStreamSynchronize(s));   // hz is reused by the next chunk (pageable memory: the copy above is staged)
    }
    if (steps > 0) w->have_flows = true;
    // results
    if (arrival || peak) {
        HIP_TRY(w->arr_out.ensure(sizeof(int32_t) * compact));
        HIP_TRY(w->peak_out.ensure(sizeof(double) * compact));
        LossArgs la{};
        la.n = n; la.npad = npad; la.count = B; la.steps = steps;
        la.arrival = ea.arrival; la.peak = ea.peak;
        la.loss = nullptr; la.arrival_out = w->arr_out.as<int32_t>(); la.peak_out = w->peak_out.as<double>();
        HIP_TRY(launch_loss(la, s));
        if (arrival) HIP_TRY(hipMemcpy(arrival, w->arr_out.p, sizeof(int32_t) * compact, hipMemcpyDeviceToHost));
        if (peak) HIP_TRY(hipMemcpy(peak, w->peak_out.p, sizeof(double) * compact, hipMemcpyDeviceToHost));
    }
    if (i_series && steps > 0)
        HIP_TRY(hipMemcpy(i_series, w->series_dev.p, sizeof(double) * (size_t)steps * compact, hipMemcpyDeviceToHost));
    if (out_states) {
        int32_t rc = world_download_states(w, out_states);
        if (rc) return rc;
    }
    return collect_errors(w->work.err.as<int32_t>(), B, 0);
}

EPI_API int32_t epi_world_keep_transitions(epi_world* w, int32_t on) {
    if (!w) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_keep_transitions: null handle");
    HIP_TRY(hipSetDevice(w->device));
    if (on) {
        const size_t cells = (size_t)w->batch * w->npad;
        HIP_TRY(w->prev_st.ensure(cells * 5 * sizeof(int32_t)));
        HIP_TRY(w->prev_cap.ensure(cells * sizeof(double)));
        HIP_TRY(w->prev_tcur.ensure(cells * sizeof(int32_t)));
        HIP_TRY(w->loc_s.ensure(cells * sizeof(int32_t)));
        HIP_TRY(w->loc_i.ensure(cells * sizeof(int32_t)));
    }
    w->keep_tr = on != 0;
    w->have_tr = false;
    return EPI_OK;
}

namespace epi {
int32_t world_transition_view(epi_world* w, WorldTransitionView* v) {
    if (!w || !v) return fail(EPI_ERR_BAD_ARGUMENT, "world_transition_view: null argument");
    if (!w->keep_tr || !w->have_tr)
        return fail(EPI_ERR_BAD_ARGUMENT, "the world holds no transition: call epi_world_keep_transitions(w, 1) and step it first");
    v->device = w->device; v->n = w->n; v->npad = w->npad; v->batch = w->batch; v->t_max = w->base.t_max;
    v->plane = (size_t)w->batch * w->npad;
    v->prev_st = w->prev_st.as<int32_t>(); v->prev_cap = w->prev_cap.as<double>(); v->prev_tcur = w->prev_tcur.as<int32_t>();
    v->loc_s = w->loc_s.as<int32_t>(); v->loc_i = w->loc_i.as<int32_t>();
    v->st = w->work.st.as<int32_t>(); v->cap = w->work.cap.as<double>();
    v->reward = w->reward_dev.as<double>(); v->act = w->act_dev.as<int8_t>();
    return EPI_OK;
}
void world_transition_consumed(epi_world* w) {
    if (w) w->have_tr = false;
}
}  // namespace epi

EPI_API int32_t epi_world_screen_stats(epi_world* w, int64_t* screened, int64_t* flagged) {
    if (!w) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_screen_stats: null handle");
    HIP_TRY(hipSetDevice(w->device));
    return w->mlp.read_stats(screened, flagged);
}

EPI_API int32_t epi_grid_screen_stats(epi_grid* g, int64_t* screened, int64_t* flagged) {
    if (!g) return fail(EPI_ERR_BAD_ARGUMENT, "epi_grid_screen_stats: null handle");
    HIP_TRY(hipSetDevice(g->device));
    return g->mlp.read_stats(screened, flagged);
}

EPI_API int32_t epi_world_set_timing(epi_world* w, int32_t on) {
    if (!w) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_set_timing: null handle");
    HIP_TRY(hipSetDevice(w->device));
    if (on && !w->ev0) {
        HIP_TRY(hipEventCreate(&w->ev0));
        HIP_TRY(hipEventCreate(&w->ev1));
    }
    w->timing = on != 0;
    if (on) { w->kernel_ms = 0.0; w->kernel_launches = 0; w->kernel_country_steps = 0; }
    return EPI_OK;
}

EPI_API int32_t epi_world_kernel_time(epi_world* w, double* total_ms, int64_t* launches, int64_t* country_steps) {
    if (!w) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_kernel_time: null handle");
    if (total_ms) *total_ms = w->kernel_ms;
    if (launches) *launches = w->kernel_launches;
    if (country_steps) *country_steps = w->kernel_country_steps;
    return EPI_OK;
}

EPI_API int32_t epi_world_last_flows(epi_world* w, int32_t* moved_s, int32_t* moved_i) {
    if (!w || !moved_s || !moved_i) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_last_flows: null argument");
    if (!w->have_flows) return fail(EPI_ERR_BAD_ARGUMENT, "epi_world_last_flows: no step has run yet");
    HIP_TRY(hipSetDevice(w->device));
    const int n = w->n, npad = w->npad, B = w->batch;
    const size_t cells = (size_t)B * npad;
    std::vector<int32_t> a(cells), b2(cells);
    HIP_TRY(hipMemcpy(a.data(), w->flow_s.p, cells * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(b2.data(), w->flow_i.p, cells * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < n; ++c) {
            moved_s[(size_t)b * n + c] = a[(size_t)b * npad + c];
            moved_i[(size_t)b * n + c] = b2[(size_t)b * npad + c];
        }
    return EPI_OK;
}
