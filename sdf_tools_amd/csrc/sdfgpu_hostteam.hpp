// sdfgpu_hostteam.hpp -- the host-side concurrency of the library as plain C++ (no HIP, no RCCL): every device call sits behind a
// callback, so that the schedulers themselves run under -fsanitize=thread in the CPU suite (tests/sched_harness.cpp, VERDICT r5
// "next round" 5) exactly as the policy objects of sdfgpu_policy.hpp run under ASan / UBSan.
//
//   HostTeam        worker threads that live as long as their context, parked on a condition variable between jobs (rounds 2 - 5
//                   spawned up to 32 std::threads PER CALL of a host-buffer entry point)
//   staged_upload   pageable host memory -> device through two pinned staging chunks: the team fills one chunk (a memcpy, or the
//                   cells / mask -> bits classification) while the DMA of the other is in flight
//   staged_drain    the mirror image: the DMA lands in one chunk while the team copies the other out into (possibly untouched)
//                   host memory
//   RankTeam        libsdfgpu_multi's one-thread-per-rank step dispatcher, with the any-rank-failed agreement in front of
//                   every exchange step and a watchdog for the exchange that was already posted
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace sdfgpu {

// ---- HostTeam -------------------------------------------------------------------------------------------------------------------
// run(n, fn): fn(0) on the calling thread, fn(1) ... fn(n - 1) on workers; returns when all have returned.  One job at a time (a
// context is used by one host thread at a time: include/sdfgpu.h).  Workers are created on demand and never go away before the team.
class HostTeam {
public:
    HostTeam() = default;
    HostTeam(const HostTeam&) = delete;
    HostTeam& operator=(const HostTeam&) = delete;
    ~HostTeam() {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (std::thread& t : th_) t.join();
    }
    int workers() const { return (int)th_.size(); }
    template <class F>
    void run(int n, F&& fn) {
        if (n <= 1) { if (n == 1) fn(0); return; }
        std::function<void(int)> job = [&fn](int w) { fn(w); };
        while ((int)th_.size() < n - 1) {
            const int idx = (int)th_.size() + 1;
            th_.emplace_back([this, idx] { worker(idx); });
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job;
            active_ = n;
            pending_ = n - 1;
            ++epoch_;
        }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    void worker(int idx) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* job = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return quit_ || (epoch_ != seen && idx < active_); });
                if (quit_) return;
                seen = epoch_;
                job = job_;
            }
            (*job)(idx);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_cv_.notify_one();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    uint64_t epoch_ = 0;
    int active_ = 0, pending_ = 0;
    bool quit_ = false;
    const std::function<void(int)>* job_ = nullptr;
};

// how many team members a transfer gets: a quarter of the host's hardware threads, capped (the fills are memory-bound: more
// threads than memory channels buy nothing and the rank threads of libsdfgpu_multi each bring a team of their own)
inline int host_team_size(int cap) {
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max<size_t>(1, std::min<size_t>((size_t)cap, hw / 4));
}

// ---- staged_upload --------------------------------------------------------------------------------------------------------------
//   fill(buf, offset_in_chunk, offset_in_upload, len)   produce bytes [offset_in_upload, + len) into staging buffer `buf` (0 / 1)
//   issue(chunk, buf, len) -> int                       enqueue the DMA of that buffer's first len bytes to the device and record
//                                                       the buffer's event; 0 or an error code
//   wait(buf) -> int                                    wait for the buffer's event
// Slices are multiples of 4096 bytes except the last.  Team member 0 is the calling thread: it fills slice 0 of the first chunk
// (a one-chunk upload -- 16 MiB of bits for 512^3 -- would otherwise only wait), then issues and waits; its slice of the later
// chunks goes to one more member.  kPerChunkCounters = false restores the arithmetic of rounds 2 - 5a (ONE running total of
// filled slices while two chunks may be in flight: the fast members' slices of chunk i + 1 stand in for a slow member's slice
// of chunk i, and chunk i goes to the device unfinished) -- kept only so that the harness can show the tool catching it.
template <bool kPerChunkCounters = true, class Fill, class Issue, class Wait>
int staged_upload(HostTeam& team_pool, size_t bytes, size_t chunk, int team, Fill&& fill, Issue&& issue, Wait&& wait) {
    if (bytes == 0) return 0;
    const int64_t nchunks = (int64_t)((bytes + chunk - 1) / chunk);
    team = std::max(team, 1);
    std::atomic<int64_t> may_fill{1};                    // chunks 0 .. may_fill may be written to their staging buffer
    std::vector<std::atomic<int>> filled((size_t)nchunks);
    for (auto& f : filled) f.store(0, std::memory_order_relaxed);
    std::atomic<int64_t> filled_total{0};                // (the pre-fix arithmetic)
    std::atomic<bool> abort{false};
    auto chunk_bytes = [&](int64_t i) { return std::min(chunk, bytes - (size_t)i * chunk); };
    auto slice = [&](int64_t i, int w) {
        const size_t len = chunk_bytes(i);
        const size_t per = ((len / (size_t)team) + 4095) & ~(size_t)4095;
        const size_t b = std::min(len, (size_t)w * per), e = std::min(len, b + per);
        if (e > b) fill((int)(i & 1), b, (size_t)i * chunk + b, e - b);
        if (kPerChunkCounters) filled[(size_t)i].fetch_add(1, std::memory_order_release);
        else filled_total.fetch_add(1, std::memory_order_release);
    };
    auto wait_may_fill = [&](int64_t i) -> bool {
        while (may_fill.load(std::memory_order_acquire) < i) {
            if (abort.load(std::memory_order_relaxed)) return false;
            std::this_thread::yield();
        }
        return true;
    };
    int err = 0;
    const int members = team + (nchunks > 1 ? 1 : 0);
    team_pool.run(members, [&](int w) {
        if (w == 0) {
            slice(0, 0);
            for (int64_t i = 0; i < nchunks && err == 0; ++i) {
                if (kPerChunkCounters) while (filled[(size_t)i].load(std::memory_order_acquire) < team) std::this_thread::yield();
                else while (filled_total.load(std::memory_order_acquire) < (i + 1) * team) std::this_thread::yield();
                err = issue(i, (int)(i & 1), chunk_bytes(i));
                // chunk i + 1 was released for filling already; chunk i + 2 shares this chunk's buffer: release it once this DMA is done
                if (err == 0 && i + 2 < nchunks) {
                    err = wait((int)(i & 1));
                    may_fill.store(i + 2, std::memory_order_release);
                }
            }
            if (err != 0) abort.store(true, std::memory_order_relaxed);
        } else if (w < team) {
            for (int64_t i = 0; i < nchunks; ++i) {
                if (!wait_may_fill(i)) return;
                slice(i, w);
            }
        } else {                                        // member `team`: the caller's slice of chunks 1 ...
            for (int64_t i = 1; i < nchunks; ++i) {
                if (!wait_may_fill(i)) return;
                slice(i, 0);
            }
        }
    });
    if (err == 0) err = wait((int)((nchunks - 1) & 1));
    if (err == 0 && nchunks > 1) err = wait((int)((nchunks - 2) & 1));
    return err;
}

// ---- staged_drain ---------------------------------------------------------------------------------------------------------------
//   issue(chunk, buf, len) -> int     enqueue the DMA device -> staging buffer `buf` and record the buffer's event
//   wait(buf) -> int                  wait for it
//   drain(buf, offset_in_chunk, offset_in_download, len)   copy those bytes out of the staging buffer
// Member 0 (the caller) issues and waits; members 1 .. team copy out.
template <class Issue, class Wait, class Drain>
int staged_drain(HostTeam& team_pool, size_t bytes, size_t chunk, int team, Issue&& issue, Wait&& wait, Drain&& drain) {
    if (bytes == 0) return 0;
    const int64_t nchunks = (int64_t)((bytes + chunk - 1) / chunk);
    team = std::max(team, 1);
    std::atomic<int64_t> ready{-1};                      // highest chunk whose bytes are in its staging buffer
    std::vector<std::atomic<int>> copied((size_t)nchunks);   // slices copied out, per chunk
    for (auto& c : copied) c.store(0, std::memory_order_relaxed);
    std::atomic<bool> abort{false};
    auto chunk_bytes = [&](int64_t i) { return std::min(chunk, bytes - (size_t)i * chunk); };
    int err = 0;
    team_pool.run(team + 1, [&](int w) {
        if (w == 0) {
            err = issue(0, 0, chunk_bytes(0));
            if (err == 0 && nchunks > 1) err = issue(1, 1, chunk_bytes(1));
            for (int64_t i = 0; i < nchunks && err == 0; ++i) {
                err = wait((int)(i & 1));
                if (err != 0) break;
                ready.store(i, std::memory_order_release);
                while (copied[(size_t)i].load(std::memory_order_acquire) < team) std::this_thread::yield();      // buffer i & 1 is free again
                if (i + 2 < nchunks) err = issue(i + 2, (int)(i & 1), chunk_bytes(i + 2));
            }
            if (err != 0) abort.store(true, std::memory_order_relaxed);
        } else {
            const int s = w - 1;
            for (int64_t i = 0; i < nchunks; ++i) {
                while (ready.load(std::memory_order_acquire) < i) {
                    if (abort.load(std::memory_order_relaxed)) return;
                    std::this_thread::yield();
                }
                const size_t len = chunk_bytes(i);
                const size_t per = ((len / (size_t)team) + 4095) & ~(size_t)4095;
                const size_t b = std::min(len, (size_t)s * per), e = std::min(len, b + per);
                if (e > b) drain((int)(i & 1), b, (size_t)i * chunk + b, e - b);
                copied[(size_t)i].fetch_add(1, std::memory_order_release);
            }
        }
    });
    return err;
}

// ---- RankTeam -------------------------------------------------------------------------------------------------------------------
// One host thread per rank for the life of a multi-GPU context (rank 0 = the calling thread).  A build is a short list of STEPS
// that every rank thread executes for its own rank.  With RCCL the ranks never wait for each other on the host (message matching
// orders the devices); when several logical ranks share a GPU a host barrier stands between the step that records an event and
// the step that waits for it (barrier_after).
//
// Failure: a rank whose step fails skips its remaining steps.  Without host barriers its peers would still post
// their sends / receives to it, those never match, and the peers block for ever in their wait step.  So
//   (1) a failing rank publishes itself (first_failed) at once, and a rank about to run an EXCHANGE step first looks: if anybody
//       has failed it skips the exchange and everything behind it (code kPeerFailed) -- no barrier on the success path, one
//       relaxed load per exchange;
//   (2) that look can come too early (the peer fails after it), or a third rank may already have posted: the rank that failed
//       then waits for the others to come back, and when they do not within `stuck_timeout_ms` it calls on_stuck() ONCE -- in
//       libsdfgpu_multi: ncclCommAbort on every communicator, which ends the posted operations -- and waits again.
struct RankStep {
    std::function<int(int)> fn;     // what rank q does (0 or an error code)
    bool barrier_after = false;     // every rank must have finished this step before any starts the next (copy mode only)
    bool wait = false;              // a step that only waits for the device: not counted as host time
    bool exchange = false;          // posts messages that only complete when every peer posts its own
};

class RankTeam {
public:
    static constexpr int kPeerFailed = -1000;       // (internal: never returned to a caller; run() reports the rank that failed first)
    int G = 1;
    int stuck_timeout_ms = 2000;
    std::function<void()> on_stuck;                 // may be empty
    std::function<void(int)> thread_init;           // called once on rank q's thread (hipSetDevice)
    std::vector<int> rc;
    std::vector<double> busy_us;                    // host time of the last job per rank thread, waits and barriers excluded
    bool aborted = false;                           // on_stuck() has been called at some point

    RankTeam() = default;
    RankTeam(const RankTeam&) = delete;
    RankTeam& operator=(const RankTeam&) = delete;
    ~RankTeam() { stop(); }

    void start(int ranks) {
        G = ranks;
        rc.assign((size_t)G, 0);
        busy_us.assign((size_t)G, 0.0);
        for (int q = 1; q < G; ++q) th_.emplace_back(&RankTeam::worker, this, q);
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (std::thread& t : th_) t.join();
        th_.clear();
    }
    // every rank runs `steps`; returns the rank that failed first (or -1)
    int run(const std::vector<RankStep>& steps) {
        finished_.store(0, std::memory_order_relaxed);
        first_failed_.store(-1, std::memory_order_relaxed);
        if (G > 1) {
            {
                std::lock_guard<std::mutex> lk(m_);
                job_ = &steps;
                ++epoch_;
            }
            cv_.notify_all();
        }
        run_rank(0, steps);
        while (finished_.load(std::memory_order_acquire) < G) std::this_thread::yield();
        const int bad = first_failed_.load(std::memory_order_acquire);
        if (bad >= 0) return bad;
        for (int q = 0; q < G; ++q) if (rc[(size_t)q] != 0) return q;
        return -1;
    }

private:
    static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void barrier() {
        if (G <= 1) return;
        const int s = bar_sense_.load(std::memory_order_acquire);
        if (bar_count_.fetch_add(1, std::memory_order_acq_rel) + 1 == G) {
            bar_count_.store(0, std::memory_order_relaxed);
            bar_sense_.store(s ^ 1, std::memory_order_release);
        } else {
            while (bar_sense_.load(std::memory_order_acquire) == s) std::this_thread::yield();
        }
    }
    void run_rank(int q, const std::vector<RankStep>& steps) {
        int my = 0;
        double busy = 0.0;
        bool i_failed_first = false;
        for (const RankStep& st : steps) {
            if (my == 0 && st.exchange && first_failed_.load(std::memory_order_acquire) >= 0) my = kPeerFailed;
            if (my == 0) {                      // (a failed rank still meets the others at the barriers)
                const double t0 = now_us();
                my = st.fn(q);
                if (!st.wait) busy += now_us() - t0;
                if (my != 0) {
                    int none = -1;
                    i_failed_first = first_failed_.compare_exchange_strong(none, q, std::memory_order_acq_rel);
                }
            }
            if (st.barrier_after) barrier();
        }
        rc[(size_t)q] = my;
        busy_us[(size_t)q] = busy;
        if (i_failed_first && G > 1) {
            // the others either skip their exchanges (they saw the flag) or sit in an exchange that waits for me
            const double t0 = now_us();
            bool fired = false;
            while (finished_.load(std::memory_order_acquire) < G - 1) {
                if (!fired && on_stuck && now_us() - t0 > 1e3 * stuck_timeout_ms) {
                    fired = true;
                    aborted = true;
                    on_stuck();
                }
                std::this_thread::yield();
            }
        }
        finished_.fetch_add(1, std::memory_order_release);
    }
    void worker(int q) {
        if (thread_init) thread_init(q);
        uint64_t seen = 0;
        for (;;) {
            const std::vector<RankStep>* steps;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return quit_ || epoch_ != seen; });
                if (quit_) return;
                seen = epoch_;
                steps = job_;
            }
            run_rank(q, *steps);
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    uint64_t epoch_ = 0;            // (guarded by m_) bumped for every job
    bool quit_ = false;
    const std::vector<RankStep>* job_ = nullptr;
    std::atomic<int> finished_{0};
    std::atomic<int> first_failed_{-1};
    std::atomic<int> bar_count_{0}, bar_sense_{0};
};

}  // namespace sdfgpu
