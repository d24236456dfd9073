// sdfgpu_multi.cpp -- libsdfgpu_multi.so: the x-slab multi-GPU build behind a C ABI (include/sdfgpu_multi.h).
//
// One host thread drives n ranks (one per GPU): per rank a libsdfgpu context, a compute stream and a
// communication stream.  The schedule is the one of sdf_tools_amd/slab.py (which runs it with one process per GPU
// on torch.distributed), built from the same stage entry points of include/sdfgpu.h; the inter-GPU messages go
// through RCCL (one group of ncclSend / ncclRecv per exchange = one message per peer and direction over the
// direct xGMI links), or -- when several ranks share a GPU, which RCCL does not allow: the single-GPU test form --
// through device-to-device copies of the same messages.
// Host code only (no kernels): compiled with hipcc for the HIP / RCCL headers.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sdfgpu_multi.h"
#include "sdfgpu_hostteam.hpp"

namespace {

thread_local std::string g_multi_create_error;

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct Rank {
    int dev = 0;
    sdfgpu_handle ctx = nullptr;
    hipStream_t s = nullptr, cs = nullptr;      // compute / communication stream
    hipEvent_t ev_s = nullptr, ev_cs = nullptr;
    ncclComm_t comm = nullptr;
    int64_t x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    Buf mask, out, cells, bits, ext, lines, out_y, sendbuf, recvtmp;
    uint32_t* d_small = nullptr;                // [0] max d^2 free, [1] filled, [2] status, [3] uncertified, [4] far hint
    uint32_t* h_small = nullptr;                // pinned
    std::string error;                          // message of this rank's last failure (written by its own host thread)
};

struct Msg {
    int src, dst;
    const void* sp;
    void* dp;
    size_t bytes;
};

}  // namespace

struct sdfgpu_multi_context {
    std::vector<Rank> r;
    bool use_rccl = false;
    bool rccl_dead = false;         // the communicators were aborted after a rank failed with an exchange in flight: the context is
                                    // finished (every later build returns an error; destroy and create again)
    std::string error;
    int halo = 3;
    bool dense_on = true;
    int last_path = 0;
    // General path without a host decision in the middle (round 4).  Which x sweep a general build runs -- halo planes +
    // slab-local sweep, or the re-partition to complete lines -- used to be decided by reading every rank's far hint back
    // after the z / y sweeps (one host round trip with all GPUs idle), and a second read found out whether the halo had
    // been enough.  RCCL sends cannot be guarded by a device word, so the choice is now PREDICTED from the previous
    // general build and validated by the ONE read at the end that the synchronous API needs anyway (maxima, completion):
    // a wrong "near" costs the re-partition on top (exact either way), a wrong "far" costs nothing but the next prediction.
    bool predict_far = false;
    int whole_hold = 0;             // builds left that keep the whole-line sweep after a halo sweep came back unresolved
    int host_reads = 0;             // read_small round trips of the last build
    int mispredictions = 0;         // general builds (since creation) whose predicted x sweep had to be redone
    // dense tier: after a failed attempt the next `dense_skip` builds leave it out (retry cadence dense_retry, doubling)
    int dense_retry = 15, dense_skip = 0, dense_fail_streak = 0;
    double host_us_max = 0.0, host_us_sum = 0.0;   // host time of the last build: slowest rank thread / sum over the rank threads
    struct sdfgpu_multi_team* team = nullptr;       // one host thread per rank (rank 0 = the caller)
};

namespace {

int mfail(sdfgpu_multi_handle h, int code, const char* fmt, ...) {
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->error = buf; else g_multi_create_error = buf;
    return code;
}

#define M_HIP(h, expr)                                                                                       \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return mfail(h, SDFGPU_ERR_HIP, "HIP error %d (%s) at %s", (int)e_, hipGetErrorString(e_), #expr); \
    } while (0)
#define M_NCCL(h, expr)                                                                                      \
    do {                                                                                                     \
        ncclResult_t e_ = (expr);                                                                            \
        if (e_ != ncclSuccess)                                                                               \
            return mfail(h, SDFGPU_ERR_HIP, "RCCL error %d (%s) at %s", (int)e_, ncclGetErrorString(e_), #expr); \
    } while (0)
#define M_SDF(h, rk, expr)                                                                                   \
    do {                                                                                                     \
        int rc_ = (expr);                                                                                    \
        if (rc_ != SDFGPU_OK) return mfail(h, rc_, "rank %d: %s", (int)(rk), sdfgpu_last_error((h)->r[rk].ctx)); \
    } while (0)

int ensure(sdfgpu_multi_handle h, Rank& k, Buf& b, size_t bytes) {
    if (b.p && b.bytes >= bytes) return SDFGPU_OK;
    M_HIP(h, hipSetDevice(k.dev));
    // (through the rank's libsdfgpu context: in red-zone mode -- SDFGPU_REDZONE=1 -- these buffers carry canaries like the context's
    //  own, and every stage call of that context checks them)
    const int q = (int)(&k - h->r.data());
    if (b.p) { M_SDF(h, q, sdfgpu_device_free(k.ctx, b.p)); b.p = nullptr; b.bytes = 0; }
    M_SDF(h, q, sdfgpu_device_malloc(k.ctx, bytes, &b.p));
    b.bytes = bytes;
    return SDFGPU_OK;
}

void slab_range(int64_t n, int rank, int world, int64_t* a, int64_t* b) {
    *a = (rank * n) / world;
    *b = ((rank + 1) * n) / world;
}

bool dense_shape_ok(int64_t nz) {
    const int64_t nzw = nz / 32;
    return (nz % 32) == 0 && nzw >= 1 && nzw <= 64 && (nzw & (nzw - 1)) == 0;
}

// ---- the rank team (round 5, VERDICT r4 "next round" 2) -----------------------------------------------------------------------
// Rounds 2 - 4 issued every rank's launches, event calls and ncclSend / ncclRecv from ONE host thread, rank after rank: ~50 - 57 us
// of host calls per extra rank (profiles/r04_bench_slab_world1.json: dense build 0.176 / 0.242 / 0.576 ms at 1 / 2 / 8 logical
// ranks against ~0.14 ms of GPU work per rank at 1024^3 / 8) -- host-bound at 8 GPUs by its own numbers.  Now every rank has
// its own host thread for the lifetime of the context (rank 0 is the calling thread), the model of one process per GPU that
// slab.py runs on torch.distributed: a build is a short list of STEPS that every rank thread executes for its own rank --
// its launches on its own streams, its own ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on its own communicator -- so
// the host cost of a build is the cost of ONE rank's calls whatever the rank count.  With RCCL the ranks never wait for each
// other on the host (message matching orders the devices); when several logical ranks share a GPU (the single-GPU test form:
// device-to-device copies that read the SENDER's buffer behind the sender's event) a host barrier stands between the step that
// records an event and the step that waits for it.
// The dispatcher itself is host-only code (sdfgpu_hostteam.hpp: RankTeam, run under -fsanitize=thread by tests/sched_harness.cpp).
// A rank whose step fails skips its remaining steps; the others look before every EXCHANGE step and skip it too, and when one of
// them had posted already the failing rank's watchdog aborts the communicators (abort_comms) so that the build returns the
// error instead of hanging in hipStreamSynchronize.
using Step = sdfgpu::RankStep;
using Team = sdfgpu::RankTeam;

int rfail(Rank& k, int code, const char* fmt, ...) {
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    k.error = buf;
    return code;
}
#define R_HIP(k, expr)                                                                                          \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) return rfail(k, SDFGPU_ERR_HIP, "HIP error %d (%s) at %s", (int)e_, hipGetErrorString(e_), #expr); \
    } while (0)
#define R_SDF(k, expr)                                                                                          \
    do {                                                                                                        \
        int rc_ = (expr);                                                                                       \
        if (rc_ != SDFGPU_OK) return rfail(k, rc_, "%s", sdfgpu_last_error((k).ctx));                             \
    } while (0)

}  // namespace

struct sdfgpu_multi_team { Team t; };

namespace {

Team& team_of(sdfgpu_multi_handle h) { return h->team->t; }

int run_steps(sdfgpu_multi_handle h, const std::vector<Step>& steps) {
    Team& t = team_of(h);
    (void)hipSetDevice(h->r[0].dev);                        // (rank 0 runs on the calling thread)
    const int bad = t.run(steps);
    double mx = 0.0, sum = 0.0;
    for (double v : t.busy_us) { mx = std::max(mx, v); sum += v; }
    h->host_us_max += mx;
    h->host_us_sum += sum;
    if (t.aborted && h->use_rccl) h->rccl_dead = true;
    if (bad >= 0) return mfail(h, t.rc[(size_t)bad], "rank %d: %s%s", bad, h->r[(size_t)bad].error.c_str(),
                               h->rccl_dead ? " (peers had posted an exchange: the RCCL communicators were aborted, destroy this context)" : "");
    return SDFGPU_OK;
}

// rank q: my communication stream waits for the compute work enqueued so far (RCCL: my own; copies: every rank's, because a
// copy reads the sender's buffer) / my compute stream waits for the exchange
// (neighbours: the exchange is between x neighbours only -- bit-plane and int32 halos -- so in copy mode a rank waits for ranks
//  q - 1, q, q + 1 instead of all of them; the re-partition is all-to-all)
int cs_after_s(sdfgpu_multi_handle h, int q, bool neighbours = false) {
    Rank& k = h->r[(size_t)q];
    const int G = (int)h->r.size();
    if (h->use_rccl) { R_HIP(k, hipStreamWaitEvent(k.cs, k.ev_s, 0)); }
    else for (int o = neighbours ? std::max(q - 1, 0) : 0; o <= (neighbours ? std::min(q + 1, G - 1) : G - 1); ++o)
        R_HIP(k, hipStreamWaitEvent(k.cs, h->r[(size_t)o].ev_s, 0));
    return SDFGPU_OK;
}
int s_after_cs(sdfgpu_multi_handle h, int q, bool neighbours = false) {
    Rank& k = h->r[(size_t)q];
    const int G = (int)h->r.size();
    if (h->use_rccl) { R_HIP(k, hipStreamWaitEvent(k.s, k.ev_cs, 0)); }
    else for (int o = neighbours ? std::max(q - 1, 0) : 0; o <= (neighbours ? std::min(q + 1, G - 1) : G - 1); ++o)
        R_HIP(k, hipStreamWaitEvent(k.s, h->r[(size_t)o].ev_cs, 0));
    return SDFGPU_OK;
}

// rank q's part of one exchange, on its communication stream: with RCCL its own sends and receives inside its own group (the
// peers post theirs from their threads: one message per peer and direction over the direct xGMI links); otherwise the copies
// it RECEIVES.
int exchange_rank(sdfgpu_multi_handle h, int q, const std::vector<Msg>& msgs) {
    Rank& k = h->r[(size_t)q];
    if (h->use_rccl) {
        bool any = false;
        for (const Msg& m : msgs) any |= m.bytes != 0 && (m.src == q || m.dst == q);
        if (!any) return SDFGPU_OK;
        ncclResult_t e = ncclGroupStart();
        if (e != ncclSuccess) return rfail(k, SDFGPU_ERR_HIP, "RCCL error %d (%s) at ncclGroupStart", (int)e, ncclGetErrorString(e));
        for (const Msg& m : msgs) {             // (an error inside the group must still close it: an open group swallows every later RCCL call)
            if (m.bytes == 0 || e != ncclSuccess) continue;
            if (m.src == q) e = ncclSend(m.sp, m.bytes, ncclChar, m.dst, k.comm, k.cs);
            if (e == ncclSuccess && m.dst == q) e = ncclRecv(m.dp, m.bytes, ncclChar, m.src, k.comm, k.cs);
        }
        const ncclResult_t e2 = ncclGroupEnd();
        if (e != ncclSuccess) return rfail(k, SDFGPU_ERR_HIP, "RCCL error %d (%s) in ncclSend / ncclRecv", (int)e, ncclGetErrorString(e));
        if (e2 != ncclSuccess) return rfail(k, SDFGPU_ERR_HIP, "RCCL error %d (%s) at ncclGroupEnd", (int)e2, ncclGetErrorString(e2));
    } else {
        for (const Msg& m : msgs)
            if (m.bytes != 0 && m.dst == q) R_HIP(k, hipMemcpyAsync(m.dp, m.sp, m.bytes, hipMemcpyDeviceToDevice, k.cs));
    }
    return SDFGPU_OK;
}

// the status block of rank q -> its pinned host copy (enqueued), and the wait for it
int read_small_enqueue(sdfgpu_multi_handle h, int q) {
    Rank& k = h->r[(size_t)q];
    R_HIP(k, hipMemcpyAsync(k.h_small, k.d_small, 32, hipMemcpyDeviceToHost, k.s));
    return SDFGPU_OK;
}
int wait_rank(sdfgpu_multi_handle h, int q) {
    Rank& k = h->r[(size_t)q];
    R_HIP(k, hipStreamSynchronize(k.s));
    return SDFGPU_OK;
}

// The limits of sdfgpu_build* (check_dims in sdfgpu.hip), applied before anything is allocated or uploaded: an oversized
// or malformed request must come back as UNSUPPORTED_SIZE / INVALID_ARGUMENT, not as a hipMalloc failure half-way.
int check_request(sdfgpu_multi_handle h, int64_t nx, int64_t ny, int64_t nz, const void* cells, size_t stride, size_t off) {
    const int G = (int)h->r.size();
    if (nx <= 0 || ny <= 0 || nz <= 0) return mfail(h, SDFGPU_ERR_INVALID_ARGUMENT, "grid dimensions must be positive");
    constexpr int64_t kMaxDim = 16384;
    if (nx > kMaxDim || ny > kMaxDim || nz > kMaxDim || nx * nx + ny * ny + nz * nz >= (1ll << 30))
        return mfail(h, SDFGPU_ERR_UNSUPPORTED_SIZE, "grid %lld x %lld x %lld exceeds the supported extent (dims <= 16384, nx^2+ny^2+nz^2 < 2^30)",
                     (long long)nx, (long long)ny, (long long)nz);
    if (nx < G) return mfail(h, SDFGPU_ERR_INVALID_ARGUMENT, "grid has fewer x planes (%lld) than ranks (%d)", (long long)nx, G);
    if (cells && (stride < 4 || (stride % 4) || (off % 4) || off + 4 > stride))
        return mfail(h, SDFGPU_ERR_INVALID_ARGUMENT, "cell_stride/occupancy_offset must be 4-byte aligned and in range");
    return SDFGPU_OK;
}

int build_device(sdfgpu_multi_handle h, const uint8_t* const* d_mask, int64_t nx, int64_t ny, int64_t nz, double res, int vb,
                 float* const* d_out, double* out_max, double* out_min) {
    const int G = (int)h->r.size();
    if (h->rccl_dead) return mfail(h, SDFGPU_ERR_HIP, "this context's RCCL communicators were aborted after a rank failed mid-exchange; destroy it and create a new one");
    if (int rc = check_request(h, nx, ny, nz, nullptr, 0, 0)) return rc;
    const int64_t plane = ny * nz;
    int64_t min_slab = nx;
    for (int q = 0; q < G; ++q) {
        Rank& k = h->r[q];
        slab_range(nx, q, G, &k.x0, &k.x1);
        slab_range(ny, q, G, &k.y0, &k.y1);
        min_slab = std::min(min_slab, k.x1 - k.x0);
        if (!d_mask[q] || !d_out[q]) return mfail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null slab pointer for rank %d", q);
    }
    h->last_path = h->use_rccl ? 4 : 0;
    h->host_reads = 0;
    h->host_us_max = h->host_us_sum = 0.0;
    uint32_t max_f = 0, max_q = 0;
    const bool host_barriers = !h->use_rccl && G > 1;     // ranks that share a GPU exchange by copies behind each other's events

    // ---- dense tier: pack -> 2 bit-planes per neighbour -> ball kernel -----------------------------------------------
    // (skipped while the handle's recent dense attempts failed: a far-field stream used to pay a dense attempt -- pack,
    //  ball kernel up to its early-out, one host read -- in front of every build; retried after `dense_retry` builds, the
    //  pause doubling while the attempts keep failing, like the single-GPU policy)
    const int64_t hb = 2;
    bool dense = h->dense_on && !vb && dense_shape_ok(nz) && min_slab >= hb;
    if (dense && h->dense_skip > 0) { --h->dense_skip; dense = false; }
    bool need_general = true;
    if (dense) {
        const int64_t wplane = ny * (nz / 32);
        for (int q = 0; q < G; ++q) {
            Rank& k = h->r[q];
            const int64_t hl = q > 0 ? hb : 0, hh = q < G - 1 ? hb : 0, nxs = k.x1 - k.x0;
            if (int rc = ensure(h, k, k.bits, (size_t)(hl + nxs + hh) * wplane * 4)) return rc;
        }
        std::vector<Msg> msgs;
        for (int q = 0; q + 1 < G; ++q) {       // boundary planes between rank q and q + 1, both directions
            Rank& a = h->r[q];
            Rank& b = h->r[q + 1];
            const int64_t al = q > 0 ? hb : 0, an = a.x1 - a.x0;
            uint32_t* abits = (uint32_t*)a.bits.p;
            uint32_t* bbits = (uint32_t*)b.bits.p;
            msgs.push_back({q, q + 1, abits + (al + an - hb) * wplane, bbits, (size_t)hb * wplane * 4});
            msgs.push_back({q + 1, q, bbits + hb * wplane, abits + (al + an) * wplane, (size_t)hb * wplane * 4});
        }
        auto phase = [&](int q, int ph) -> int {
            Rank& k = h->r[(size_t)q];
            const int64_t hl = q > 0 ? hb : 0, hh = q < G - 1 ? hb : 0;
            R_SDF(k, sdfgpu_slab_dense_phase(k.ctx, ph, d_mask[q], k.x1 - k.x0, ny, nz, (uint32_t*)k.bits.p, hl, hh, res, d_out[q],
                                             k.d_small, k.s));
            return SDFGPU_OK;
        };
        std::vector<Step> steps;
        steps.push_back({[&](int q) -> int {                    // boundary planes first: they are what the neighbours wait for
            Rank& k = h->r[(size_t)q];
            if (int rc = phase(q, 0)) return rc;
            if (G > 1) R_HIP(k, hipEventRecord(k.ev_s, k.s));
            return SDFGPU_OK; }, host_barriers, false});
        steps.push_back({[&](int q) -> int {                    // exchange on the communication stream, the interior while the messages fly
            Rank& k = h->r[(size_t)q];
            if (G > 1) {
                if (int rc = cs_after_s(h, q, true)) return rc;
                if (int rc = exchange_rank(h, q, msgs)) return rc;
            }
            if (int rc = phase(q, 1)) return rc;
            if (G > 1) R_HIP(k, hipEventRecord(k.ev_cs, k.cs));
            return SDFGPU_OK; }, host_barriers, false, true});
        steps.push_back({[&](int q) -> int {                    // the 2 + 2 border planes, then the status block
            if (G > 1) if (int rc = s_after_cs(h, q, true)) return rc;
            if (int rc = phase(q, 2)) return rc;
            return read_small_enqueue(h, q); }, false, false});
        steps.push_back({[&](int q) -> int { return wait_rank(h, q); }, false, true});
        ++h->host_reads;
        if (int rc = run_steps(h, steps)) return rc;
        need_general = false;
        for (Rank& k : h->r) {
            need_general |= k.h_small[3] != 0;
            max_f = std::max(max_f, k.h_small[0]);
            max_q = std::max(max_q, k.h_small[1]);
        }
        if (!need_general) { h->last_path |= 1; h->dense_fail_streak = 0; }
        else {
            h->dense_fail_streak = std::min(h->dense_fail_streak + 1, 5);
            h->dense_skip = h->dense_retry > 0 ? std::min(h->dense_retry << (h->dense_fail_streak - 1), 255) : 0;
        }
    }

    if (need_general) {
        const int64_t halo = std::max<int64_t>(0, std::min<int64_t>(h->halo, min_slab));
        for (int q = 0; q < G; ++q) {
            Rank& k = h->r[q];
            const int64_t hl = q > 0 ? halo : 0, hh = q < G - 1 ? halo : 0, nxs = k.x1 - k.x0;
            if (int rc = ensure(h, k, k.ext, (size_t)(hl + nxs + hh) * plane * 4)) return rc;
        }
        // ---- slab-local z / y sweeps (tier picked on the device), far hint ------------------------------------------
        // (no host read behind them: the far hints stay in d_small[4] and come back with the final status block)
        auto zy_step = [&](int q) -> int {
            Rank& k = h->r[(size_t)q];
            const int64_t hl = q > 0 ? halo : 0;
            R_HIP(k, hipMemsetAsync(k.d_small, 0, 32, k.s));
            R_SDF(k, sdfgpu_sweep_zy_tiered_device(k.ctx, d_mask[q], k.x1 - k.x0, ny, nz, (int32_t*)k.ext.p + hl * plane, k.d_small + 4, k.s));
            return SDFGPU_OK;
        };
        // One rank holds complete lines already: its "re-partition" is the identity, and the whole-line sweep -- exact on any
        // scene, near- or far-field chosen on the device -- IS its x sweep (round 4 ran a halo sweep, two 0.5 GB self-copies and
        // the line sweep there: 1.66 ms against 0.855 for the single-GPU ABI on the two-box scene).
        bool far = G == 1 || h->predict_far || h->whole_hold > 0;
        bool hinted = false, zy_done = false;
        if (!far) {
            // ---- near-field: `halo` int32 planes per neighbour, x sweep that reports voxels needing more ----------------
            std::vector<Msg> msgs;
            for (int q = 0; q + 1 < G && halo > 0; ++q) {
                Rank& a = h->r[q];
                Rank& b = h->r[q + 1];
                const int64_t al = q > 0 ? halo : 0, an = a.x1 - a.x0;
                int32_t* ae = (int32_t*)a.ext.p;
                int32_t* be = (int32_t*)b.ext.p;
                msgs.push_back({q, q + 1, ae + (al + an - halo) * plane, be, (size_t)halo * plane * 4});
                msgs.push_back({q + 1, q, be + halo * plane, ae + (al + an) * plane, (size_t)halo * plane * 4});
            }
            std::vector<Step> steps;
            steps.push_back({[&](int q) -> int {
                Rank& k = h->r[(size_t)q];
                if (int rc = zy_step(q)) return rc;
                R_HIP(k, hipEventRecord(k.ev_s, k.s));
                return SDFGPU_OK; }, host_barriers, false});
            steps.push_back({[&](int q) -> int {
                Rank& k = h->r[(size_t)q];
                if (int rc = cs_after_s(h, q, true)) return rc;
                if (int rc = exchange_rank(h, q, msgs)) return rc;
                R_HIP(k, hipEventRecord(k.ev_cs, k.cs));
                return SDFGPU_OK; }, host_barriers, false, true});
            steps.push_back({[&](int q) -> int {
                Rank& k = h->r[(size_t)q];
                if (int rc = s_after_cs(h, q, true)) return rc;
                const int64_t hl = q > 0 ? halo : 0, hh = q < G - 1 ? halo : 0, nxs = k.x1 - k.x0;
                R_SDF(k, sdfgpu_sweep_x_device(k.ctx, (const int32_t*)k.ext.p, hl, nxs, hh, ny, nz, k.x0 - hl > 0, k.x1 + hh < nx, k.x0,
                                               nx, res, vb, d_out[q], k.d_small, k.d_small + 2, k.s));
                return read_small_enqueue(h, q); }, false, false});          // the build's one host round trip when the prediction holds
            steps.push_back({[&](int q) -> int { return wait_rank(h, q); }, false, true});
            ++h->host_reads;
            if (int rc = run_steps(h, steps)) return rc;
            zy_done = true;
            max_f = max_q = 0;
            bool unresolved = false;
            for (Rank& k : h->r) {
                unresolved |= k.h_small[2] != 0;
                hinted |= k.h_small[4] != 0;
                max_f = std::max(max_f, k.h_small[0]);
                max_q = std::max(max_q, k.h_small[1]);
            }
            // Only an UNRESOLVED voxel forces the redo: with every voxel resolved inside the halo the result is exact already,
            // whatever the hint says -- the hint then only steers the next build's prediction.
            far = unresolved;
            h->predict_far = hinted;
            if (far) {
                ++h->mispredictions;
                if (!hinted) h->whole_hold = 8;                 // near-field clutter with a cavity deeper than the halo:
            }                                                   // stay on complete lines for a while instead of flapping
        } else if (h->whole_hold > 0) --h->whole_hold;
        if (far && G == 1) {
            h->last_path |= 2;
            std::vector<Step> steps;
            steps.push_back({[&](int q) -> int {
                Rank& k = h->r[(size_t)q];
                if (!zy_done) if (int rc = zy_step(q)) return rc;
                R_HIP(k, hipMemsetAsync(k.d_small, 0, 16, k.s));      // (maxima, status; word 4 keeps the far hint)
                R_SDF(k, sdfgpu_sweep_x_lines_device(k.ctx, (const int32_t*)k.ext.p, nx, ny, nz, 0, ny, res, vb, d_out[q], k.d_small, k.s));
                return read_small_enqueue(h, q); }, false, false});
            steps.push_back({[&](int q) -> int { return wait_rank(h, q); }, false, true});
            ++h->host_reads;
            if (int rc = run_steps(h, steps)) return rc;
            max_f = h->r[0].h_small[0];
            max_q = h->r[0].h_small[1];
            h->predict_far = h->r[0].h_small[4] != 0;
        } else if (far) {
            // ---- far-field: x slabs -> y slabs, exact x sweep on complete lines, back to x slabs ------------------------
            h->last_path |= 2;
            for (int q = 0; q < G; ++q) {
                Rank& k = h->r[q];
                const int64_t nxs = k.x1 - k.x0, nys = k.y1 - k.y0;
                if (int rc = ensure(h, k, k.sendbuf, (size_t)nxs * plane * 4)) return rc;
                if (int rc = ensure(h, k, k.recvtmp, (size_t)nxs * plane * 4)) return rc;
                if (int rc = ensure(h, k, k.lines, (size_t)nx * std::max<int64_t>(nys, 1) * nz * 4)) return rc;
                if (int rc = ensure(h, k, k.out_y, (size_t)nx * std::max<int64_t>(nys, 1) * nz * 4)) return rc;
            }
            std::vector<Msg> there, back;
            for (int q = 0; q < G; ++q) {       // there: my rows of rank d's y slab, contiguous per destination, into rows [x0_q, x1_q) of its line buffer
                Rank& k = h->r[q];
                const int64_t nxs = k.x1 - k.x0, nys = k.y1 - k.y0;
                for (int d = 0; d < G; ++d) {
                    Rank& o = h->r[d];
                    const int64_t dys = o.y1 - o.y0, dxs = o.x1 - o.x0;
                    if (d == q) continue;
                    if (dys > 0) there.push_back({q, d, (int32_t*)k.sendbuf.p + nxs * o.y0 * nz, (int32_t*)o.lines.p + k.x0 * dys * nz, (size_t)nxs * dys * nz * 4});
                    // back: rows [x0_d, x1_d) of my y slab (contiguous) -> rank d, which scatters them into its x slab
                    if (nys > 0) back.push_back({q, d, (const float*)k.out_y.p + o.x0 * nys * nz, (float*)o.recvtmp.p + dxs * k.y0 * nz, (size_t)dxs * nys * nz * 4});
                }
            }
            std::vector<Step> steps;
            steps.push_back({[&](int q) -> int {                // pack: strided -> contiguous per destination; my own y slab straight into my line buffer
                Rank& k = h->r[(size_t)q];
                if (!zy_done) if (int rc = zy_step(q)) return rc;
                const int64_t hl = q > 0 ? halo : 0, nxs = k.x1 - k.x0;
                const int32_t* own = (const int32_t*)k.ext.p + hl * plane;
                for (int d = 0; d < G; ++d) {
                    Rank& o = h->r[(size_t)d];
                    const int64_t dys = o.y1 - o.y0;
                    if (dys == 0) continue;
                    const size_t width = (size_t)dys * nz * 4;
                    int32_t* dst = d == q ? (int32_t*)k.lines.p + k.x0 * dys * nz : (int32_t*)k.sendbuf.p + nxs * o.y0 * nz;
                    R_HIP(k, hipMemcpy2DAsync(dst, width, own + o.y0 * nz, (size_t)plane * 4, width, (size_t)nxs, hipMemcpyDeviceToDevice, k.s));
                }
                R_HIP(k, hipEventRecord(k.ev_s, k.s));
                return SDFGPU_OK; }, host_barriers, false});
            steps.push_back({[&](int q) -> int {
                Rank& k = h->r[(size_t)q];
                if (int rc = cs_after_s(h, q)) return rc;
                if (int rc = exchange_rank(h, q, there)) return rc;
                R_HIP(k, hipEventRecord(k.ev_cs, k.cs));
                return SDFGPU_OK; }, host_barriers, false, true});
            steps.push_back({[&](int q) -> int {
                Rank& k = h->r[(size_t)q];
                if (int rc = s_after_cs(h, q)) return rc;
                const int64_t nys = k.y1 - k.y0, dxs = k.x1 - k.x0;
                R_HIP(k, hipMemsetAsync(k.d_small, 0, 16, k.s));      // (maxima, status; word 4 keeps the far hint)
                if (nys > 0) {
                    R_SDF(k, sdfgpu_sweep_x_lines_device(k.ctx, (const int32_t*)k.lines.p, nx, nys, nz, k.y0, ny, res, vb,
                                                         (float*)k.out_y.p, k.d_small, k.s));
                    // my own rows of my y slab go straight into my x slab
                    R_HIP(k, hipMemcpy2DAsync(d_out[q] + k.y0 * nz, (size_t)plane * 4, (const float*)k.out_y.p + k.x0 * nys * nz, (size_t)nys * nz * 4,
                                              (size_t)nys * nz * 4, (size_t)dxs, hipMemcpyDeviceToDevice, k.s));
                }
                R_HIP(k, hipEventRecord(k.ev_s, k.s));
                return SDFGPU_OK; }, host_barriers, false});
            steps.push_back({[&](int q) -> int {
                Rank& k = h->r[(size_t)q];
                if (int rc = cs_after_s(h, q)) return rc;
                if (int rc = exchange_rank(h, q, back)) return rc;
                R_HIP(k, hipEventRecord(k.ev_cs, k.cs));
                return SDFGPU_OK; }, host_barriers, false, true});
            steps.push_back({[&](int q) -> int {                // unpack what I received
                Rank& k = h->r[(size_t)q];
                if (int rc = s_after_cs(h, q)) return rc;
                const int64_t dxs = k.x1 - k.x0;
                for (const Msg& m : back) {
                    if (m.dst != q || m.bytes == 0) continue;
                    Rank& src = h->r[(size_t)m.src];
                    const int64_t nys = src.y1 - src.y0;
                    R_HIP(k, hipMemcpy2DAsync(d_out[q] + src.y0 * nz, (size_t)plane * 4, m.dp, (size_t)nys * nz * 4, (size_t)nys * nz * 4,
                                              (size_t)dxs, hipMemcpyDeviceToDevice, k.s));
                }
                return read_small_enqueue(h, q); }, false, false});
            steps.push_back({[&](int q) -> int { return wait_rank(h, q); }, false, true});
            ++h->host_reads;
            if (int rc = run_steps(h, steps)) return rc;
            max_f = max_q = 0;
            hinted = false;
            for (Rank& k : h->r) {
                max_f = std::max(max_f, k.h_small[0]);
                max_q = std::max(max_q, k.h_small[1]);
                hinted |= k.h_small[4] != 0;
            }
            // (the whole-line sweep is exact on any scene: a stale "far" costs nothing but this update)
            h->predict_far = hinted;
        }
    }
    // (every rank's wait step synchronised its compute stream; the communication streams are ordered before them)
    // red-zone mode: one more look at every rank's buffers, the exchanged ones included (a no-op otherwise)
    for (int q = 0; q < G; ++q) {
        M_HIP(h, hipSetDevice(h->r[(size_t)q].dev));
        M_SDF(h, q, sdfgpu_redzone_check(h->r[(size_t)q].ctx, h->r[(size_t)q].s));
    }
    M_HIP(h, hipSetDevice(h->r[0].dev));
    return sdfgpu_extrema_from_dsq(max_f, max_q, res, out_max, out_min);
}

int build_host(sdfgpu_multi_handle h, const uint8_t* filled, const void* cells, size_t stride, size_t off, int unknown,
               int64_t nx, int64_t ny, int64_t nz, double res, int vb, float* out, double* out_max, double* out_min) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if ((!filled && !cells) || !out) return mfail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null host pointer");
    const int G = (int)h->r.size();
    if (int rc = check_request(h, nx, ny, nz, cells, stride, off)) return rc;
    const int64_t plane = ny * nz;
    std::vector<const uint8_t*> dm((size_t)G);
    std::vector<float*> dout((size_t)G);
    for (int q = 0; q < G; ++q) {
        Rank& k = h->r[q];
        slab_range(nx, q, G, &k.x0, &k.x1);
        const int64_t nxs = k.x1 - k.x0;
        if (int rc = ensure(h, k, k.mask, (size_t)nxs * plane)) return rc;
        if (int rc = ensure(h, k, k.out, (size_t)nxs * plane * 4)) return rc;
        dm[(size_t)q] = (const uint8_t*)k.mask.p;
        dout[(size_t)q] = (float*)k.out.p;
    }
    // Every rank has its own link to the host: the slabs go up (and come back) side by side, each driven by its rank's host
    // thread.  Up: sdfgpu_upload_classified -- the rank's thread team classifies its slab of the caller's mask / cells into one
    // bit per voxel while it fills the pinned staging chunks (1/8 B per voxel over the link), a kernel spreads the bits into
    // the rank's byte mask.  Down: sdfgpu_copy_to_host (pinned chunks drained by the team into the caller's possibly untouched
    // memory).  Ranks that share a GPU share its link: they go one after the other.
    bool shared_gpu = false;
    for (int q = 0; q < G; ++q) for (int p = 0; p < q; ++p) shared_gpu |= h->r[(size_t)p].dev == h->r[(size_t)q].dev;
    std::mutex link;
    std::vector<Step> up, down;
    up.push_back({[&](int q) -> int {
        Rank& k = h->r[(size_t)q];
        const int64_t nxs = k.x1 - k.x0;
        std::unique_lock<std::mutex> lk(link, std::defer_lock);
        if (shared_gpu) lk.lock();
        R_SDF(k, sdfgpu_upload_classified(k.ctx, cells ? nullptr : filled + (size_t)k.x0 * plane,
                                          cells ? (const char*)cells + (size_t)k.x0 * plane * stride : nullptr, stride, off, unknown,
                                          nxs * plane, (uint8_t*)k.mask.p, k.s));
        return SDFGPU_OK; }, false, true});
    down.push_back({[&](int q) -> int {
        Rank& k = h->r[(size_t)q];
        std::unique_lock<std::mutex> lk(link, std::defer_lock);
        if (shared_gpu) lk.lock();
        R_SDF(k, sdfgpu_copy_to_host(k.ctx, out + (size_t)k.x0 * plane, k.out.p, (size_t)(k.x1 - k.x0) * plane * 4, k.s));
        return SDFGPU_OK; }, false, true});
    if (int rc = run_steps(h, up)) return rc;
    if (int rc = build_device(h, dm.data(), nx, ny, nz, res, vb, dout.data(), out_max, out_min)) return rc;
    const double us_max = h->host_us_max, us_sum = h->host_us_sum;      // (the build's own host time, not the download's)
    if (int rc = run_steps(h, down)) return rc;
    h->host_us_max = us_max; h->host_us_sum = us_sum;
    return SDFGPU_OK;
}

}  // namespace

extern "C" {

int sdfgpu_multi_create(int n_ranks, const int* devices, sdfgpu_multi_handle* out_handle) {
    if (!out_handle) return mfail(nullptr, SDFGPU_ERR_INVALID_ARGUMENT, "out_handle is null");
    *out_handle = nullptr;
    if (n_ranks < 1 || n_ranks > 64) return mfail(nullptr, SDFGPU_ERR_INVALID_ARGUMENT, "n_ranks must be 1..64");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return mfail(nullptr, SDFGPU_ERR_NO_DEVICE, "no HIP device available; this library has no CPU fallback");
    sdfgpu_multi_context* h = new (std::nothrow) sdfgpu_multi_context();
    if (!h) return mfail(nullptr, SDFGPU_ERR_INVALID_ARGUMENT, "out of host memory");
    h->r.resize((size_t)n_ranks);
    std::set<int> distinct;
    std::vector<int> devs((size_t)n_ranks);
    for (int q = 0; q < n_ranks; ++q) {
        devs[(size_t)q] = devices ? devices[q] : q;
        if (devs[(size_t)q] < 0 || devs[(size_t)q] >= ndev) {
            delete h;
            return mfail(nullptr, SDFGPU_ERR_NO_DEVICE, "rank %d: device index %d out of range (0..%d)", q, devs[(size_t)q], ndev - 1);
        }
        distinct.insert(devs[(size_t)q]);
    }
    h->use_rccl = (int)distinct.size() == n_ranks;          // RCCL: one rank per device
    auto bail = [&](int code, const std::string& msg) {
        g_multi_create_error = msg;
        sdfgpu_multi_destroy(h);
        return code;
    };
    for (int q = 0; q < n_ranks; ++q) {
        Rank& k = h->r[(size_t)q];
        k.dev = devs[(size_t)q];
        if (hipSetDevice(k.dev) != hipSuccess) return bail(SDFGPU_ERR_HIP, "hipSetDevice failed");
        const int rc = sdfgpu_create(k.dev, &k.ctx);
        if (rc != SDFGPU_OK) return bail(rc, std::string("sdfgpu_create: ") + sdfgpu_last_error(nullptr));
        if (hipStreamCreateWithFlags(&k.s, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&k.cs, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&k.ev_s, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&k.ev_cs, hipEventDisableTiming) != hipSuccess ||
            hipMalloc((void**)&k.d_small, 64) != hipSuccess || hipMemset(k.d_small, 0, 64) != hipSuccess ||
            hipHostMalloc((void**)&k.h_small, 64, hipHostMallocDefault) != hipSuccess)
            return bail(SDFGPU_ERR_HIP, "HIP stream / event / status-block allocation failed");
    }
    if (h->use_rccl) {
        std::vector<ncclComm_t> comms((size_t)n_ranks);
        const ncclResult_t e = ncclCommInitAll(comms.data(), n_ranks, devs.data());
        if (e != ncclSuccess) return bail(SDFGPU_ERR_HIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(e));
        for (int q = 0; q < n_ranks; ++q) h->r[(size_t)q].comm = comms[(size_t)q];
        // peers exchange device buffers directly: make sure peer access is on where the runtime needs it asked for
        for (int a = 0; a < n_ranks; ++a)
            for (int b = 0; b < n_ranks; ++b) {
                int can = 0;
                if (a != b && hipDeviceCanAccessPeer(&can, devs[(size_t)a], devs[(size_t)b]) == hipSuccess && can) {
                    (void)hipSetDevice(devs[(size_t)a]);
                    (void)hipDeviceEnablePeerAccess(devs[(size_t)b], 0);   // "already enabled" is fine
                    (void)hipGetLastError();
                }
            }
    }
    h->team = new (std::nothrow) sdfgpu_multi_team();
    if (!h->team) return bail(SDFGPU_ERR_INVALID_ARGUMENT, "out of host memory");
    h->team->t.thread_init = [devs](int q) { (void)hipSetDevice(devs[(size_t)q]); };   // (the current device is per thread: set once)
    if (h->use_rccl) h->team->t.on_stuck = [h]() {
        // a rank failed while peers sat in an exchange that waits for it: end the posted operations.  The communicators are
        // gone afterwards (rccl_dead): the build returns its error, every later build says why it cannot run.
        for (Rank& k : h->r) if (k.comm) { (void)ncclCommAbort(k.comm); k.comm = nullptr; }
    };
    h->team->t.start(n_ranks);                              // one host thread per rank beyond the caller's (rank 0)
    *out_handle = h;
    return SDFGPU_OK;
}

int sdfgpu_multi_destroy(sdfgpu_multi_handle h) {
    if (!h) return SDFGPU_OK;
    if (h->team) { h->team->t.stop(); delete h->team; h->team = nullptr; }
    for (Rank& k : h->r) {
        (void)hipSetDevice(k.dev);
        if (k.s) (void)hipStreamSynchronize(k.s);
        if (k.cs) (void)hipStreamSynchronize(k.cs);
        if (k.comm) (void)ncclCommDestroy(k.comm);
        for (Buf* b : {&k.mask, &k.out, &k.cells, &k.bits, &k.ext, &k.lines, &k.out_y, &k.sendbuf, &k.recvtmp})
            if (b->p && k.ctx) (void)sdfgpu_device_free(k.ctx, b->p);
        if (k.d_small) (void)hipFree(k.d_small);
        if (k.h_small) (void)hipHostFree(k.h_small);
        if (k.ev_s) (void)hipEventDestroy(k.ev_s);
        if (k.ev_cs) (void)hipEventDestroy(k.ev_cs);
        if (k.s) (void)hipStreamDestroy(k.s);
        if (k.cs) (void)hipStreamDestroy(k.cs);
        if (k.ctx) (void)sdfgpu_destroy(k.ctx);
    }
    delete h;
    return SDFGPU_OK;
}

const char* sdfgpu_multi_last_error(sdfgpu_multi_handle h) { return h ? h->error.c_str() : g_multi_create_error.c_str(); }

int sdfgpu_multi_ranks(sdfgpu_multi_handle h) { return h ? (int)h->r.size() : 0; }

int sdfgpu_multi_slab_range(sdfgpu_multi_handle h, int64_t nx, int rank, int64_t* x0, int64_t* x1) {
    if (!h || !x0 || !x1 || rank < 0 || rank >= (int)h->r.size()) return SDFGPU_ERR_INVALID_ARGUMENT;
    slab_range(nx, rank, (int)h->r.size(), x0, x1);
    return SDFGPU_OK;
}

int sdfgpu_multi_build(sdfgpu_multi_handle h, const uint8_t* filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                       int add_virtual_border, float* out_sdf, double* out_max, double* out_min) {
    return build_host(h, filled, nullptr, 0, 0, 0, nx, ny, nz, resolution, add_virtual_border, out_sdf, out_max, out_min);
}

int sdfgpu_multi_build_cells(sdfgpu_multi_handle h, const void* cells, size_t cell_stride, size_t occupancy_offset,
                             int unknown_is_filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                             int add_virtual_border, float* out_sdf, double* out_max, double* out_min) {
    if (h && !cells) return mfail(h, SDFGPU_ERR_INVALID_ARGUMENT, "cells is null");
    return build_host(h, nullptr, cells, cell_stride, occupancy_offset, unknown_is_filled, nx, ny, nz, resolution,
                      add_virtual_border, out_sdf, out_max, out_min);
}

int sdfgpu_multi_build_device(sdfgpu_multi_handle h, const uint8_t* const* d_mask_slabs, int64_t nx, int64_t ny, int64_t nz,
                              double resolution, int add_virtual_border, float* const* d_out_slabs, double* out_max,
                              double* out_min) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!d_mask_slabs || !d_out_slabs) return mfail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null pointer array");
    // the caller's buffers may have been produced on other streams: settle them first
    for (Rank& k : h->r) {
        M_HIP(h, hipSetDevice(k.dev));
        M_HIP(h, hipDeviceSynchronize());
    }
    return build_device(h, d_mask_slabs, nx, ny, nz, resolution, add_virtual_border, d_out_slabs, out_max, out_min);
}

int sdfgpu_multi_last_path(sdfgpu_multi_handle h, int* out_bits) {
    if (!h || !out_bits) return SDFGPU_ERR_INVALID_ARGUMENT;
    *out_bits = h->last_path;
    return SDFGPU_OK;
}

int sdfgpu_multi_last_stats(sdfgpu_multi_handle h, int* out_host_reads, int* out_mispredictions) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (out_host_reads) *out_host_reads = h->host_reads;
    if (out_mispredictions) *out_mispredictions = h->mispredictions;
    return SDFGPU_OK;
}

int sdfgpu_multi_last_host_us(sdfgpu_multi_handle h, double* out_max_rank_us, double* out_sum_us) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (out_max_rank_us) *out_max_rank_us = h->host_us_max;
    if (out_sum_us) *out_sum_us = h->host_us_sum;
    return SDFGPU_OK;
}

int sdfgpu_multi_set_option(sdfgpu_multi_handle h, const char* name, int value) {
    if (!h || !name) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!strcmp(name, "dense_retry")) { h->dense_retry = std::max(0, value); h->dense_skip = 0; h->dense_fail_streak = 0; }   // (and forwarded)
    if (!strcmp(name, "halo")) { h->halo = std::max(0, value); return SDFGPU_OK; }
    if (!strcmp(name, "predict_far")) { h->predict_far = value != 0; h->whole_hold = 0; return SDFGPU_OK; }
    if (!strcmp(name, "dense")) h->dense_on = value != 0;     // (also forwarded: the ranks' own dense tier is not used here)
    for (size_t q = 0; q < h->r.size(); ++q) M_SDF(h, q, sdfgpu_set_option(h->r[q].ctx, name, value));
    return SDFGPU_OK;
}

}  // extern "C"
