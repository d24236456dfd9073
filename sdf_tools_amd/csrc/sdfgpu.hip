// sdfgpu.hip -- C ABI (include/sdfgpu.h) over the gfx950 kernels in sdfgpu_kernels.hpp.
// Host side of the drop-in boundary for sdf_generation::ExtractSignedDistanceField
// (reference include/sdf_tools/sdf_generation.hpp:209-420).  No CPU fallback.
#include "sdfgpu_kernels.hpp"
#include "sdfgpu_sweep_x16.hpp"
#include "sdfgpu_sweep_y16.hpp"
#include "sdfgpu_dense3.hpp"
#include "sdfgpu_dense6.hpp"
#include "sdfgpu_fused_zy.hpp"
#include "sdfgpu_dense.hpp"
#include "sdfgpu_envelope_dc.hpp"
#include "sdfgpu_policy.hpp"
#include "sdfgpu_hostteam.hpp"

#include <sys/mman.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sdfgpu.h"

using namespace sdfgpu;

namespace {

constexpr int kH = 3;            // register-window radius of the marching sweeps
constexpr int64_t kMaxDim = 16384;

thread_local std::string g_create_error = "";

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct sdfgpu_context {
    int device = 0;
    std::string error;
    DeviceBuffer zfield;     // int16 [N]   K1 output
    DeviceBuffer yzfield;    // int32 [N]   K2 output (32-bit pipeline) / side table (16-bit pipeline)
    DeviceBuffer plane16;    // int16 [N]   plane field of the 16-bit pipeline
    DeviceBuffer bits;       // uint32 [N/32] packed occupancy of the dense path
    DeviceBuffer unc;        // uint32 [N/32] undecided bits handed from the dense ball kernel to its fix-up kernel
    DeviceBuffer tileflag;   // uint32 [tiles] which waves of a tile wrote unc words (kept zero between builds)
    DeviceBuffer fix_order;  // uint32 [kFixRows] (dx, dy) rows of the fix-up kernel sorted by dx^2 + dy^2
    DeviceBuffer tagmask;    // uint8 [N] mask produced by the tagged-object classify kernel
    DeviceBuffer tagids;     // uint32 object id filter
    DeviceBuffer stage_in;   // host-API staging: mask / cells
    DeviceBuffer stage_bits; // host-API staging: the host-classified bit field (1 bit per voxel, linear order)
    int host_pack = 1;       // host-buffer builds classify on the host and upload bits (option "host_pack"; 0 = upload the
                             // caller's mask / cells and classify on the device, as rounds 1 - 4 did)
    DeviceBuffer stage_out;  // host-API staging: sdf
    DeviceBuffer query_stage;   // host-API staging of sdfgpu_query_points: points | distance | gradient | flags
    size_t tag_cached_bytes = 0;            // stage_in holds the tagged cell records of the last sdfgpu_build_tagged_cells call
    void* pin[2] = {nullptr, nullptr};      // pinned host staging of copy_to_host (two chunks in flight)
    // Red zones (round 6, VERDICT r5 "next round" 2): with SDFGPU_REDZONE=1 in the environment of sdfgpu_create (or option
    // "redzone") every device allocation of the library -- the scratch fields, the status block, the slots, the staging buffers,
    // what sdfgpu_device_malloc hands out -- carries kRzPad canary bytes in front and behind, and every ABI call that may have
    // launched something ends with one check kernel over all of them (then synchronises): a store outside a buffer fails the
    // call that made it, with the buffer's name (SDFGPU_ERR_REDZONE).  The only net rounds 1 - 5 had was the VALUE of the
    // output, and a store past the end of a field lived in the product for three rounds.
    bool redzone = false;
    int rz_depth = 0;                       // entry points call each other: the outermost one checks
    struct Zone { std::string name; char* base; char* user; size_t bytes, total; };
    std::vector<Zone> zones;
    bool rz_table_dirty = true;
    void* rz_table = nullptr;               // device copy of {base, user bytes, total} per zone + results
    size_t rz_table_cap = 0;
    HostTeam* team = nullptr;               // the host threads that fill / drain the staging chunks: created with the first staged
                                            // transfer, parked between calls, joined by sdfgpu_destroy (sdfgpu_hostteam.hpp)
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    uint32_t* d_small = nullptr;   // [0] max d^2 free, [1] max d^2 filled, [2] status, [3] uncertified, [4] far_y, [5] far_x
    hipStream_t last_stream = nullptr;
    double last_resolution = 1.0;
    int64_t last_n = 0;
    bool have_result = false;
    bool last_fused = false;
    bool last_predicted = false;     // the last build took the far-field pair on the strength of the handle's recent builds: no probes, no marching launches
    // "this handle's scene is a far-field scene on both axes": learnt from the status blocks that come back (never a wait).  Such a
    // build skips the two probes and the three guarded marching launches (35 us of a 0.8 - 1.4 ms build) and enqueues KE2 -> KE3
    // directly (int32 hand-off); every 16th build probes again.  The far-field pair is exact on any scene: a wrong prediction costs
    // time, never a voxel.  Option "far_predict" (0 off, 1 on, 2 every build: tests and the fuzz).
    FarHabit far;                    // (sdfgpu_policy.hpp: plain C++, driven on the CPU by tests/policy_harness.cpp)
    uint32_t* h_far = nullptr;       // second report slot (pinned): builds that do not carry the dense tier report their far flags here
    uint32_t* h_far_dev = nullptr;
    hipEvent_t far_ev = nullptr;
    bool far_pending = false;
    bool last_standby = false;       // the last build carried the far-field stand-by pair behind a trusted dense tier
    bool last_dense3 = false, last_staged = false;   // ... ran KD3 + KF in KD's place / carried them behind KD, guarded on its verdict
    int tune_ty = 0, tune_tx = 0, tune_tzy = 0, fused_h = 0;
    bool fused_zy = true;            // use K12 (z sweep fused into the y sweep) when the shape allows
    bool dense_on = true;            // try the bit-parallel dense kernel first when the shape allows
    int pack_variant = 0;
    int ball_block = 0;
    DensePolicy pol;              // what the handle has learned about its scenes' dense tier (sdfgpu_policy.hpp)
    uint32_t* unc_override = nullptr;   // set around the staged KD3 launch: undecided bits in the z field's storage
    size_t unc_override_bytes = 0;
    int defer_fold = 0;           // stage entry points leave their maxima in the slot array until sdfgpu_fold_extrema_device
    int ball_variant = 0;         // debugging: bit0 = bounds-checked expansion, bit1 = generic (non-ZINV) expansion
    int nt_store = 0;               // measured: non-temporal output stores slow the next build's pack (0.03 -> 0.08 ms)
    bool envelope_on = true;         // bound the outward scans and redo far-field sweeps with the envelope kernels
    bool envelope_dc = true;         // use the divide-and-conquer envelope kernel (sdfgpu_envelope_dc.hpp) when the shape allows
    bool dense_generic_on = true;    // generic dense kernels for shapes / modes the tuned ones do not take
    bool shell_on = true;            // KD6, the bit-parallel shell pass between KD3 and KF (option "dense_shell")
    bool dc_fixed = true;            // far-field kernel: the 512-voxel-line instances (option "dc_fixed")
    bool plane_skip = true;          // builds that go straight to the far-field pair: the z sweep marks the x-planes that hold a filled voxel,
                                     // the y sweep skips the tiles of the others, the x sweep their row loads (option "plane_skip")
    bool flat_score_reset = false;   // "policy_reset": the next build clears the device-side habit of the two-valued tiles (status word 40)
    int flat_tiles = 1;              // (2: every such tile tries, whatever the habit says -- tests)
    // ... and y tiles whose 16 lines are all flat-positive (a floor under open space) skip the search (option "flat_tiles")
    DeviceBuffer planebits;          // row_any [nx * ny] bytes | plane_any [nx] bytes | row_bits [nx][ceil(ny / 32)] words (k_sweep_z_wave16, k_pack_row_flags, EnvDcArgs)
    bool last_plane_skip = false;    // the last build used it (sdfgpu_debug_copy_yzsweep fills the skipped planes in)
    int64_t last_dims[3] = {0, 0, 0};
    bool dense3_fixed = true;        // KD3's nz = 512 instance (option "dense3_fixed")
    int shell_min_words = kShellMinWords;   // option "shell_min_words"
    int shell_budget_den = 8;        // ... for scenes with at most 1 / 8 of their voxels undecided behind KD3 (option "shell_budget_den"): Bernoulli
                                     // p = 0.01 leaves 8 % (0.78 ms against the far-field pair's 0.96), p = 0.007 17 % (1.07 ms against 0.97)
    int force_env = -1;              // -1 automatic, 1 = envelope kernels only (option "envelope_mode")
    // an axis is far-field when more than 1 / den of the probed voxels have d^2 >= thr.  Per axis, from the measured
    // break-even of the two sweeps at 512^3 (tools/p_sweep.py, tools/ythr_probe.py): the y marching sweep (2 B rows, radius-8
    // windows) against the far-field PAIR (int32 hand-off, no x probe) breaks even at in-plane d^2 ~ 36 on an eighth of the
    // voxels (Bernoulli p = 0.02: 1.31 vs 1.29 ms; p = 0.015: 1.54 vs 1.29 ms; p = 0.025: 1.21 vs 1.28 ms); the x marching sweep falls behind the far-field kernel as soon as its
    // radius-3 window stops deciding nearly every voxel (Bernoulli p = 0.03: 0.65 ms against 0.55; p = 0.04 and denser:
    // marching wins), so its threshold is d^2 >= 9 on 1 / 24 of the voxels (p = 0.03: 6 %, p = 0.04: 2.4 %)
    // (round 3, third-generation far-field kernel: KE2 0.57 -> 0.39 ms moves the y break-even from Bernoulli p ~ 0.018 up to
    //  p ~ 0.035 -- tools/tier_sweep.py: p = 0.03 1.01 vs 0.97 ms, p = 0.02 1.16 vs 0.985, p = 0.04 0.87 vs 0.97 -- i.e. from
    //  "d^2 >= 36 on an eighth" to "d^2 >= 16 on a fifth" of the voxels: (1 - p)^49 = 0.225 at p = 0.03, 0.135 at p = 0.04)
    int far_thr[2] = {16, 9};
    int far_den[2] = {5, 24};
    bool i32_handoff = true;         // far-field pair: int32 plane field between the y and x sweeps (option "i32_handoff")
    bool standby_far = true;         // stand-by pipeline behind a trusted dense tier = the far-field pair (bounded whatever the scene
                                     // turns into), not fused K12 + K3/16 with unbounded scans (option "standby_far")
    bool standby_fold = true;        // the stand-by x sweep's launch folds the maxima and publishes the status block itself (option "standby_fold")
    int standby_grid = 1024;         // workgroups of the stand-by launches (LOOP form; option "standby_grid"): 4 per CU = all resident at once
    bool dc_attr_set[16] = {};   // MaxDynamicSharedMemorySize raised, per far-field kernel instantiation
    int k1_resident = 0;             // workgroups of k_sweep_z_vec16 the device holds at once (persistent grid size)
    unsigned long long* d_clocks = nullptr;   // SDFGPU_PHASE_CLOCKS builds: phase clocks of the far-field kernels
    int dc_debug = 0;                // profiling aid: skips phases of k_envelope_dc (results are then wrong)
    int dc_debug_stage = 0;          // ... of this stage only (2 / 3; 0 = both)
    uint32_t* far_y = nullptr;       // set while a build enqueues a bounded K2 / K3
    int scan_y = kScanExpectNear, scan_x = kScanExpectNear;   // outward-scan bounds of the marching kernels
    bool fused_always = false;
    uint32_t* d_slots = nullptr;     // [kSlots][kSlotWords] extrema / flag slots of the dense kernel (kept zero between launches)
    uint32_t* d_result = nullptr;    // [8] the status block of the last finished build (d_small is cleared by its fold)
    bool small_clean = false;        // d_small is all zero (left so by the previous build's fold kernel)
    uint32_t* h_flags_dev = nullptr; // device-side address of h_flags
    uint32_t* h_flags = nullptr;     // pinned host copy of the status block, written by the fold kernel of every build
    hipEvent_t flags_ev = nullptr;
    bool flags_pending = false;
    hipEvent_t build_done_ev = nullptr;   // recorded behind every build (and every stage call that uses the status block):
                                          // work on another stream waits for it first
    bool order_valid = false;             // build_done_ev has been recorded on order_stream
    hipStream_t order_stream = nullptr;
    bool last_dense = false;
    const uint32_t* guard = nullptr; // set while a build enqueues the flag-guarded general pipeline
    bool plane16_on = true;          // use the int16 plane field + side table when the shape allows
    bool z_wave_on = true;           // z sweep with wave-private rows where nz allows it (option "z_wave"; 0 = the workgroup form)
    bool probe_window = true;        // tier probes as window statistics (option "probe_window"; 0 = level A of the far-field search on sampled tiles)
    bool y16_on = true;              // y sweep of that pipeline through the packed 16-bit kernel (option "y16"; 0 = the 32-bit marching kernel)
    int x16_v = 4, x16_h = 3;        // K3/16 variant: voxels per lane, window radius
    int march_h = 3;                 // K2 (y sweep) register-window radius: 3 or 8 (forced)
    int mid_thr_y = 16, mid_den_y = 24;   // y probe: radius-8 marching window when more than 1 / den of the voxels have d^2 >= thr
    bool last_plane16 = false;
    int profiling = 0;            // 0 off, 1 an event behind every stage, 2 events around the dense ball kernel only,
                                  // 3 like 2 but only on every 4th build
    uint32_t profiled_builds = 0; // builds seen while profiling (level 3 samples every 4th)
    std::vector<hipEvent_t> event_pool;   // recycled profiling events
    std::vector<hipEvent_t> events;   // 8 per profiled build: start, after pack, ball, K1, K2/K12, KE2, K3, KE3
};

namespace {

int fail(sdfgpu_handle h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->error = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                  \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(h, SDFGPU_ERR_HIP, "HIP error %d (%s) at %s", (int)e_,            \
                        hipGetErrorString(e_), #expr);                                    \
    } while (0)

constexpr size_t kRzPad = 4096;
constexpr int kRzByte = 0xC5;

// every device allocation of the library goes through here (name = what a red-zone report calls it)
int rz_malloc(sdfgpu_handle h, const char* name, size_t bytes, void** out) {
    *out = nullptr;
    if (!h->redzone) {
        HIP_TRY(h, hipMalloc(out, bytes));
        return SDFGPU_OK;
    }
    const size_t total = kRzPad + ((bytes + 255) & ~(size_t)255) + kRzPad;
    char* base = nullptr;
    HIP_TRY(h, hipMalloc((void**)&base, total));
    HIP_TRY(h, hipMemset(base, kRzByte, kRzPad));
    HIP_TRY(h, hipMemset(base + kRzPad + bytes, kRzByte, total - kRzPad - bytes));      // (from the first byte behind the buffer)
    h->zones.push_back({name, base, base + kRzPad, bytes, total});
    h->rz_table_dirty = true;
    *out = base + kRzPad;
    return SDFGPU_OK;
}
int rz_free(sdfgpu_handle h, void* ptr) {
    if (!ptr) return SDFGPU_OK;
    for (size_t i = 0; i < h->zones.size(); ++i)
        if (h->zones[i].user == ptr) {
            char* base = h->zones[i].base;
            h->zones.erase(h->zones.begin() + (long)i);
            h->rz_table_dirty = true;
            HIP_TRY(h, hipFree(base));
            return SDFGPU_OK;
        }
    HIP_TRY(h, hipFree(ptr));               // (allocated before the switch went on)
    return SDFGPU_OK;
}

int ensure(sdfgpu_handle h, DeviceBuffer& b, size_t bytes, const char* name = "scratch") {
    if (b.bytes >= bytes && b.ptr) return SDFGPU_OK;
    if (b.ptr) { if (int rc = rz_free(h, b.ptr)) return rc; b.ptr = nullptr; b.bytes = 0; }
    // (red zones: the buffer is exactly as large as asked for, so that the first byte behind it is canary)
    const size_t want = h->redzone ? std::max<size_t>(bytes, 4) : std::max<size_t>(bytes, 256);
    if (int rc = rz_malloc(h, name, want, &b.ptr)) return rc;
    b.bytes = want;
    return SDFGPU_OK;
}

__global__ __launch_bounds__(256) void k_redzone_check(const uint64_t* __restrict__ table, uint32_t* __restrict__ result) {
    // table[3 z] = base address, [3 z + 1] = user bytes, [3 z + 2] = total bytes; result[2 z] = overwritten bytes, [2 z + 1] = first
    // overwritten byte as an offset from the buffer's start + kRzPad (so that "before the buffer" stays non-negative)
    const int z = blockIdx.x;
    const unsigned char* base = reinterpret_cast<const unsigned char*>(table[3 * z]);
    const size_t bytes = table[3 * z + 1], total = table[3 * z + 2];
    __shared__ uint32_t cnt, first;
    if (threadIdx.x == 0) { cnt = 0u; first = 0xFFFFFFFFu; }
    __syncthreads();
    uint32_t c = 0, f = 0xFFFFFFFFu;
    for (size_t i = threadIdx.x; i < kRzPad; i += 256) if (base[i] != (unsigned char)kRzByte) { ++c; f = min(f, (uint32_t)i); }
    for (size_t i = kRzPad + bytes + threadIdx.x; i < total; i += 256)
        if (base[i] != (unsigned char)kRzByte) { ++c; f = min(f, (uint32_t)min(i - bytes, (size_t)0xFFFFFFFEu)); }
    if (c) { atomicAdd(&cnt, c); atomicMin(&first, f); }
    __syncthreads();
    if (threadIdx.x == 0) { result[2 * z] = cnt; result[2 * z + 1] = first; }
}

// One kernel over every zone, then a synchronisation (debug mode: calls become synchronous).  Overwritten zones fail the call
// with the buffer's name and are repaired, so that the next call reports only what IT did.
int redzone_check(sdfgpu_handle h, hipStream_t s) {
    const size_t nz = h->zones.size();
    if (!h->redzone || nz == 0) return SDFGPU_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t need = nz * 3 * 8 + nz * 2 * 4;
    if (h->rz_table_cap < need) {
        if (h->rz_table) HIP_TRY(h, hipFree(h->rz_table));
        h->rz_table = nullptr;
        h->rz_table_cap = 0;
        HIP_TRY(h, hipMalloc(&h->rz_table, need * 2));
        h->rz_table_cap = need * 2;
        h->rz_table_dirty = true;
    }
    uint32_t* d_res = reinterpret_cast<uint32_t*>(static_cast<char*>(h->rz_table) + nz * 3 * 8);
    if (h->rz_table_dirty) {
        std::vector<uint64_t> t(nz * 3);
        for (size_t i = 0; i < nz; ++i) { t[3 * i] = reinterpret_cast<uint64_t>(h->zones[i].base); t[3 * i + 1] = h->zones[i].bytes; t[3 * i + 2] = h->zones[i].total; }
        HIP_TRY(h, hipStreamSynchronize(s));
        HIP_TRY(h, hipMemcpy(h->rz_table, t.data(), nz * 3 * 8, hipMemcpyHostToDevice));
        h->rz_table_dirty = false;
    }
    hipLaunchKernelGGL(k_redzone_check, dim3((unsigned)nz), dim3(256), 0, s, (const uint64_t*)h->rz_table, d_res);
    HIP_TRY(h, hipGetLastError());
    std::vector<uint32_t> res(nz * 2);
    HIP_TRY(h, hipMemcpyAsync(res.data(), d_res, nz * 2 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    std::string msg;
    for (size_t i = 0; i < nz; ++i) {
        if (res[2 * i] == 0) continue;
        const sdfgpu_context::Zone& z = h->zones[i];
        char buf[256];
        const long long off = (long long)res[2 * i + 1] - (long long)kRzPad;
        snprintf(buf, sizeof buf, "%s'%s' (%zu bytes): %u canary bytes overwritten, the first %lld bytes %s", msg.empty() ? "" : "; ", z.name.c_str(), z.bytes,
                 res[2 * i], off < 0 ? -off : off - (long long)z.bytes, off < 0 ? "BEFORE its start" : "BEHIND its end");
        msg += buf;
        (void)hipMemset(z.base, kRzByte, kRzPad);
        (void)hipMemset(z.base + kRzPad + z.bytes, kRzByte, z.total - kRzPad - z.bytes);
    }
    if (!msg.empty()) return fail(h, SDFGPU_ERR_REDZONE, "red zone: a kernel of this call stored outside %s", msg.c_str());
    return SDFGPU_OK;
}

template <class F>
int rz_wrap(sdfgpu_handle h, void* stream, F&& body) {
    if (!h || !h->redzone) return body();
    ++h->rz_depth;
    int rc = body();
    if (--h->rz_depth == 0) {
        const int rz = redzone_check(h, (hipStream_t)stream);
        if (rz != SDFGPU_OK) rc = rz;
    }
    return rc;
}

// The EDT is symmetric under axis renaming, and a grid with singleton axes has
// the same memory layout when those axes are moved to the front (index =
// x*ny*nz + y*nz + z).  Canonicalising keeps the contiguous axis long, so 2-D
// grids (the reference's test_bindings.py case is 20x40x1) sweep coalesced.
void canonical_dims(int64_t& nx, int64_t& ny, int64_t& nz) {
    if (nz == 1) { nz = ny; ny = nx; nx = 1; }
    if (nz == 1) { nz = ny; ny = nx; nx = 1; }
    if (ny == 1) { ny = nx; nx = 1; }
}

int check_dims(sdfgpu_handle h, int64_t nx, int64_t ny, int64_t nz) {
    if (nx <= 0 || ny <= 0 || nz <= 0)
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "grid dimensions must be positive (got %lld x %lld x %lld)",
                    (long long)nx, (long long)ny, (long long)nz);
    if (nx > kMaxDim || ny > kMaxDim || nz > kMaxDim || nx * nx + ny * ny + nz * nz >= (int64_t)kInf32)
        return fail(h, SDFGPU_ERR_UNSUPPORTED_SIZE,
                    "grid %lld x %lld x %lld exceeds the supported extent (dims <= 16384, nx^2+ny^2+nz^2 < 2^30)",
                    (long long)nx, (long long)ny, (long long)nz);
    return SDFGPU_OK;
}

int rows_per_block(int nz) {
    const int W = (nz + 63) / 64;
    return std::max(1, 4096 / (W * 64));
}

// K1 launch.  cells == nullptr -> uint8 mask.
bool z_wave16_shape(const sdfgpu_context* h, int64_t nz) {
    return h->z_wave_on && (nz == 64 || nz == 128 || nz == 256 || nz == 512 || nz == 1024);
}

// d_row_any != nullptr (callers check z_wave16_shape and, for masks, the alignment first): the kernel also writes one "holds a filled voxel" byte per z row
int launch_sweep_z(sdfgpu_handle h, const uint8_t* d_mask, const void* d_cells, size_t stride, size_t off,
                   int unknown, int64_t nx, int64_t ny, int64_t nz, int16_t* d_out, hipStream_t s, const uint32_t* d_bits = nullptr,
                   uint8_t* d_row_any = nullptr) {
    const int64_t nrows = nx * ny;
    if (d_bits) {
        // bits in (round 6): the wave-private form reads its 16 voxels as one 16-bit piece of the bit field; every other shape
        // goes through the generic kernel with a bit loader
        if (h->z_wave_on && (nz == 64 || nz == 128 || nz == 256 || nz == 512 || nz == 1024)) {
            const int rw = 1024 / (int)nz;
            const int64_t ngroups = (nrows + rw - 1) / rw;
            dim3 gw((unsigned)std::min<int64_t>((ngroups + kBlock / 64 - 1) / (kBlock / 64), 8192)), block(kBlock);
            const uint8_t* b8 = reinterpret_cast<const uint8_t*>(d_bits);
            switch ((int)nz) {
                case 64: hipLaunchKernelGGL((k_sweep_z_wave16<4, true>), gw, block, 0, s, b8, d_out, nrows, h->guard, d_row_any); break;
                case 128: hipLaunchKernelGGL((k_sweep_z_wave16<8, true>), gw, block, 0, s, b8, d_out, nrows, h->guard, d_row_any); break;
                case 256: hipLaunchKernelGGL((k_sweep_z_wave16<16, true>), gw, block, 0, s, b8, d_out, nrows, h->guard, d_row_any); break;
                case 512: hipLaunchKernelGGL((k_sweep_z_wave16<32, true>), gw, block, 0, s, b8, d_out, nrows, h->guard, d_row_any); break;
                default: hipLaunchKernelGGL((k_sweep_z_wave16<64, true>), gw, block, 0, s, b8, d_out, nrows, h->guard, d_row_any); break;
            }
        } else {
            const int rpb = rows_per_block((int)nz);
            const int W = ((int)nz + 63) / 64;
            const size_t lds = (size_t)rpb * W * 8 + (size_t)rpb * 4;
            const int64_t nblocks = (nrows + rpb - 1) / rpb;
            BitsLoader ld{d_bits};
            hipLaunchKernelGGL(k_sweep_z_generic<BitsLoader>, dim3((unsigned)std::min<int64_t>(nblocks, 2048)), dim3(kBlock), lds, s, ld, d_out, nrows, (int)nz, rpb, h->guard);
        }
        HIP_TRY(h, hipGetLastError());
        return SDFGPU_OK;
    }
    const int rpb = rows_per_block((int)nz);
    const int W = ((int)nz + 63) / 64;
    const size_t lds = (size_t)rpb * W * 8 + (size_t)rpb * 4;     // bitmap + one class word per row (vec16 kernel)
    const int64_t nblocks = (nrows + rpb - 1) / rpb;
    // persistent row-group loop inside the kernel: 8 workgroups per CU are plenty, and a small grid
    // makes the guard early-exit (dense path certified) a ~2 us launch instead of ~10 us
    dim3 grid((unsigned)std::min<int64_t>(nblocks, 2048)), block(kBlock);
    if (!d_cells && (nz % 16) == 0 && (reinterpret_cast<uintptr_t>(d_mask) % 16) == 0) {
        // the persistent grid must fit the device in ONE round: at 70 VGPRs 7 workgroups are resident per CU, and a grid of 8
        // per CU left 256 workgroups to run their 16 row groups alone after the others had finished
        if (h->k1_resident <= 0) {
            int per_cu = 0, cus = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sweep_z_vec16, kBlock, lds) != hipSuccess || per_cu <= 0) per_cu = 4;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus <= 0) cus = 256;
            h->k1_resident = per_cu * cus;
        }
        grid.x = (unsigned)std::min<int64_t>(nblocks, h->k1_resident);
    }
    if (d_cells) {
        CellLoader ld{reinterpret_cast<const char*>(d_cells), (int64_t)stride, (int64_t)off, unknown};
        hipLaunchKernelGGL(k_sweep_z_generic<CellLoader>, grid, block, lds, s, ld, d_out, nrows, (int)nz, rpb, h->guard);
    } else if (h->z_wave_on && (nz == 64 || nz == 128 || nz == 256 || nz == 512 || nz == 1024) &&
               (reinterpret_cast<uintptr_t>(d_mask) % 16) == 0) {
        // whole rows per wave (16 nz-lanes): no workgroup barrier; 32 workgroups per CU's worth of persistent waves (measured
        // at 512^3: 8 per CU 0.099 ms, 16 0.093, 32 0.088, one step per wave 0.100)
        const int rw = 1024 / (int)nz;                                  // rows per wave step
        const int64_t ngroups = (nrows + rw - 1) / rw;
        dim3 gw((unsigned)std::min<int64_t>((ngroups + kBlock / 64 - 1) / (kBlock / 64), 8192));
        switch ((int)nz) {
            case 64: hipLaunchKernelGGL((k_sweep_z_wave16<4, false>), gw, block, 0, s, d_mask, d_out, nrows, h->guard, d_row_any); break;
            case 128: hipLaunchKernelGGL((k_sweep_z_wave16<8, false>), gw, block, 0, s, d_mask, d_out, nrows, h->guard, d_row_any); break;
            case 256: hipLaunchKernelGGL((k_sweep_z_wave16<16, false>), gw, block, 0, s, d_mask, d_out, nrows, h->guard, d_row_any); break;
            case 512: hipLaunchKernelGGL((k_sweep_z_wave16<32, false>), gw, block, 0, s, d_mask, d_out, nrows, h->guard, d_row_any); break;
            default: hipLaunchKernelGGL((k_sweep_z_wave16<64, false>), gw, block, 0, s, d_mask, d_out, nrows, h->guard, d_row_any); break;
        }
    } else if ((nz % 16) == 0 && (reinterpret_cast<uintptr_t>(d_mask) % 16) == 0) {
        hipLaunchKernelGGL(k_sweep_z_vec16, grid, block, lds, s, d_mask, d_out, nrows, (int)nz, rpb, h->guard);
    } else {
        MaskLoader ld{d_mask};
        hipLaunchKernelGGL(k_sweep_z_generic<MaskLoader>, grid, block, lds, s, ld, d_out, nrows, (int)nz, rpb, h->guard);
    }
    HIP_TRY(h, hipGetLastError());
    return SDFGPU_OK;
}

int pick_T(int user, int span) {
    int T = user > 0 ? user : 64;
    T = std::max(T, 2 * 8 + 1);
    return std::min(T, std::max(span, 1));
}

template <int STAGE, bool VB>
int launch_march(sdfgpu_handle h, SweepArgs a, bool vec4, hipStream_t s, int force_window = 0) {
    const int span = a.out_hi - a.out_lo;
    const int nchunks = (span + a.T - 1) / a.T;
    const int64_t nbx = (a.ncols + kBlock - 1) / kBlock;
    if (nbx > 0x7fffffffLL || nchunks > 65535) return fail(h, SDFGPU_ERR_UNSUPPORTED_SIZE, "sweep grid too large");
    dim3 grid((unsigned)nbx, (unsigned)nchunks), block(kBlock);
    bool wide = false;
    if constexpr (STAGE == 2 && !VB) {                       // radius-8 window: y sweep only
        if (vec4 && force_window != 3 && (force_window == 8 || h->march_h == 8)) {
            hipLaunchKernelGGL((k_sweep_march<2, 4, 8, false>), grid, block, 0, s, a);
            wide = true;
        }
    }
    if (wide) {}
    else if (STAGE == 2 && vec4 && a.out16 && h->y16_on && kH == 3)      // y sweep into the 16-bit plane field: packed 16-bit window
        hipLaunchKernelGGL((k_sweep_y16<3>), grid, block, 0, s, a);
    else if (vec4) hipLaunchKernelGGL((k_sweep_march<STAGE, 4, kH, VB>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_sweep_march<STAGE, 1, kH, VB>), grid, block, 0, s, a);
    HIP_TRY(h, hipGetLastError());
    return SDFGPU_OK;
}

// K2 launch: int16 z field -> int32 in-plane signed d^2
// (d_side != nullptr: write the int16 plane field to d_out and exact saturated groups to d_side;
//  requires the vec4 path)
int launch_sweep_y(sdfgpu_handle h, const int16_t* d_in, void* d_out, int32_t* d_side, int64_t nx, int64_t ny,
                   int64_t nz, hipStream_t s, int force_window = 0) {
    const bool vec4 = (nz % 4) == 0 && (reinterpret_cast<uintptr_t>(d_in) % 8) == 0 &&
                      (reinterpret_cast<uintptr_t>(d_out) % 16) == 0;
    if (d_side && !vec4) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "16-bit plane output needs nz % 4 == 0");
    const int V = vec4 ? 4 : 1;
    SweepArgs a{};
    a.in = d_in; a.out = d_out;
    a.out16 = d_side ? 1 : 0; a.side = d_side;
    a.guard = h->guard;
    if (h->far_y) { a.max_scan = h->scan_y; a.far_flag = h->far_y; }
    a.cpl = nz / V;
    a.ncols = nx * a.cpl;
    a.outer_stride = ny * nz;
    a.line_stride = nz;
    a.L = (int)ny; a.out_lo = 0; a.out_hi = (int)ny;
    a.T = pick_T(h->tune_ty, (int)ny);
    if (h->tune_ty <= 0 && vec4 && d_side && h->y16_on) {
        // K2/16: chunks of 16 batches of 7 rows while the launch still fills the device.  Measured at 512^3 (rocprofv3):
        // 56 rows 131 us, 64 rows 138, 112 rows 122, 119 / 126 rows 123 - 125, 128 rows 132 -- with a power-of-two chunk the
        // workgroups, which march in step, read addresses that differ by multiples of 64 KB only.
        const int64_t nbx = (a.ncols + kBlock - 1) / kBlock;
        for (int t : {112, 56, 28}) {
            a.T = std::min(t, (int)ny);
            if (nbx * ((ny + t - 1) / t) >= 1024) break;
        }
    }
    return launch_march<2, false>(h, a, vec4, s, force_window);
}

// K12 launch: mask -> int32 in-plane signed d^2 in one kernel (nz = 512 or 1024, 16-B aligned mask)
bool fused_zy_eligible(const sdfgpu_context* h, const uint8_t* d_mask, const void* d_out, int64_t nz) {
    return h->fused_zy && d_mask && (nz == 512 || nz == 1024) &&
           (reinterpret_cast<uintptr_t>(d_mask) % 16) == 0 && (reinterpret_cast<uintptr_t>(d_out) % 16) == 0;
}

template <int V, int H>
void launch_fused_variant(const FusedZyArgs& a0, int T, int64_t nx, int64_t ny, bool out16, hipStream_t s) {
    constexpr int R = 2 * H + 1;
    FusedZyArgs a = a0;
    a.T = std::min(std::max(T, R), (int)ny);
    const int wpb = kBlock / 64;
    dim3 grid((unsigned)((nx + wpb - 1) / wpb), (unsigned)((ny + a.T - 1) / a.T)), block(kBlock);
    const size_t lds = (size_t)wpb * (R + 1) * (8 + 64 * (V / 8) + 8);
    if (out16) hipLaunchKernelGGL((k_sweep_zy_fused<V, H, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((k_sweep_zy_fused<V, H, false>), grid, block, lds, s, a);
}

// d_side != nullptr: d_out is the int16 plane field, saturated groups go to d_side
int launch_sweep_zy_fused(sdfgpu_handle h, const uint8_t* d_mask, void* d_out, int32_t* d_side, int64_t nx,
                          int64_t ny, int64_t nz, hipStream_t s) {
    FusedZyArgs a{};
    a.mask = d_mask; a.out = d_out; a.side = d_side; a.nx = (int)nx; a.ny = (int)ny;
    a.guard = h->guard;
    const int T = h->tune_tzy > 0 ? h->tune_tzy : 64;
    const bool o16 = d_side != nullptr;
    if (nz == 512 && h->fused_h == 3) launch_fused_variant<8, 3>(a, T, nx, ny, o16, s);
    else if (nz == 512) launch_fused_variant<8, 2>(a, T, nx, ny, o16, s);
    else launch_fused_variant<16, 2>(a, T, nx, ny, o16, s);
    HIP_TRY(h, hipGetLastError());
    return SDFGPU_OK;
}

// K3/16 launch: int16 plane field + side table (optionally with x halo) -> fp32 sdf
template <int V, int H>
void launch_x16_variant(const SweepX16Args& a, dim3 grid, dim3 block, bool vb, bool slab, hipStream_t s) {
    if (vb && slab) hipLaunchKernelGGL((k_sweep_x16<V, H, true, true>), grid, block, 0, s, a);
    else if (vb) hipLaunchKernelGGL((k_sweep_x16<V, H, true, false>), grid, block, 0, s, a);
    else if (slab) hipLaunchKernelGGL((k_sweep_x16<V, H, false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_sweep_x16<V, H, false, false>), grid, block, 0, s, a);
}

bool plane16_eligible(const sdfgpu_context* h, int64_t ny, int64_t nz) {
    return h->plane16_on && (nz % 4) == 0 && ((ny * nz) % 8) == 0;
}

int launch_sweep_x16(sdfgpu_handle h, const int16_t* d_in16, const int32_t* d_side, float* d_out, int64_t halo_lo,
                     int64_t nxs, int64_t halo_hi, int64_t side_lo, int64_t side_hi, int64_t ny, int64_t nz,
                     int lo_trunc, int hi_trunc, int64_t x_global, int64_t nx_global, double resolution, int vb,
                     uint32_t* d_maxdsq, uint32_t* d_status, hipStream_t s) {
    const int64_t plane = ny * nz;
    const int V = (h->x16_v == 8) ? 8 : 4;
    SweepX16Args a{};
    a.in16 = d_in16; a.side32 = d_side; a.out = d_out;
    a.ncols = plane / V; a.plane = plane;
    a.L = (int)(halo_lo + nxs + halo_hi);
    a.out_lo = (int)halo_lo; a.out_hi = (int)(halo_lo + nxs);
    a.T = pick_T(h->tune_tx, (int)nxs);
    a.side_lo = (int)side_lo; a.side_hi = (int)side_hi;
    a.resolution = resolution;
    a.lo_truncated = lo_trunc; a.hi_truncated = hi_trunc;
    a.x_global = x_global; a.nx_global = nx_global; a.ny = ny; a.nz = nz;
    (void)d_maxdsq;                             // maxima go to the slot array; the caller folds them (fold_slots)
    a.maxdsq = h->d_slots; a.status = d_status;
    a.guard = h->guard;
    if (h->far_y) { a.max_scan = h->scan_x; a.far_flag = h->far_y + 1; }
    const int span = a.out_hi - a.out_lo;
    const int nchunks = (span + a.T - 1) / a.T;
    const int64_t nbx = (a.ncols + kBlock - 1) / kBlock;
    if (nbx > 0x7fffffffLL || nchunks > 65535) return fail(h, SDFGPU_ERR_UNSUPPORTED_SIZE, "sweep grid too large");
    dim3 grid((unsigned)nbx, (unsigned)nchunks), block(kBlock);
    const bool slab = lo_trunc || hi_trunc || side_lo > 0 || side_hi < a.L;
    if (V == 8) launch_x16_variant<8, 3>(a, grid, block, vb != 0, slab, s);
    else if (h->x16_h == 2) launch_x16_variant<4, 2>(a, grid, block, vb != 0, slab, s);
    else if (h->x16_h == 8 && !vb && !slab)      // wide window (option only): plain grids
        hipLaunchKernelGGL((k_sweep_x16<4, 8, false, false>), grid, block, 0, s, a);
    else launch_x16_variant<4, 3>(a, grid, block, vb != 0, slab, s);
    HIP_TRY(h, hipGetLastError());
    return SDFGPU_OK;
}

// K3 launch: int32 plane field (optionally with x halo) -> fp32 sdf
int launch_sweep_x(sdfgpu_handle h, const int32_t* d_in, float* d_out, int64_t halo_lo, int64_t nxs,
                   int64_t halo_hi, int64_t ny, int64_t nz, int lo_trunc, int hi_trunc, int64_t x_global,
                   int64_t nx_global, double resolution, int vb, uint32_t* d_maxdsq, uint32_t* d_status,
                   hipStream_t s, int64_t y_off = 0, int64_t ny_glob = -1, int max_scan = 0, uint32_t* far_flag = nullptr) {
    const int64_t plane = ny * nz;
    const bool vec4 = (plane % 4) == 0 && (reinterpret_cast<uintptr_t>(d_in) % 16) == 0 &&
                      (reinterpret_cast<uintptr_t>(d_out) % 16) == 0;
    const int V = vec4 ? 4 : 1;
    SweepArgs a{};
    a.in = d_in; a.out = d_out;
    a.ncols = plane / V;
    a.cpl = a.ncols;
    a.outer_stride = 0;
    a.line_stride = plane;
    a.L = (int)(halo_lo + nxs + halo_hi);
    a.out_lo = (int)halo_lo; a.out_hi = (int)(halo_lo + nxs);
    a.T = pick_T(h->tune_tx, (int)nxs);
    a.resolution = resolution;
    a.lo_truncated = lo_trunc; a.hi_truncated = hi_trunc;
    a.x_global = x_global; a.nx_global = nx_global; a.ny = ny; a.nz = nz;
    a.y_off = y_off; a.ny_glob = ny_glob < 0 ? ny : ny_glob;
    a.max_scan = max_scan; a.far_flag = far_flag;
    (void)d_maxdsq;                             // maxima go to the slot array; the caller folds them (fold_slots)
    a.maxdsq = h->d_slots; a.status = d_status;
    a.guard = h->guard;
    return vb ? launch_march<3, true>(h, a, vec4, s) : launch_march<3, false>(h, a, vec4, s);
}

// Shapes the far-field kernel takes (any line count; keys must fit 32 bits: finf + (L + 2)^2 < 2^(32 - B)).
struct DcGeometry { bool ok; int B; uint32_t finf; int pitch; int M, Kp; };
// dynamic LDS a far-field launch may ask for: the CU's 160 KB less the kernel's few static words (the fold of the stand-by x sweep)
constexpr size_t kDcMaxDynamicLds = 160 * 1024 - 256;
struct DcExtra {                 // int32 plane fields instead of p16 + side table, y-slab geometry
    const int32_t* in_i32 = nullptr;
    int32_t* out_i32 = nullptr;
    int64_t y_off = 0, ny_glob = -1;
    const uint32_t* i32_flag = nullptr;     // device word: use the int32 fields only when it is non-zero (nullptr: always)
    bool loop = false;                      // LOOP form of the kernel: a small grid whose workgroups walk the tiles (stand-by launches)
    const uint32_t* bits = nullptr;         // stage 2: z distances from the dense tier's bit field instead of the z field (forces the scalar form)
    int nzw = 0;
    uint32_t* ran_flag = nullptr;           // status word raised by a launch that does work
    const uint32_t* row_bits = nullptr;     // plane sparsity (see EnvDcArgs); nullptr: every plane is processed
    int row_words = 0;
    const uint8_t* plane_any = nullptr;
    uint32_t* fold_result = nullptr;        // stage 3, LOOP form: the launch also does the end-of-build fold (fold_ticket = a free status word)
    uint32_t* fold_report = nullptr;
    uint32_t* fold_ticket = nullptr;
};
struct DcDecide {                // a probe launch turns its counters into the tier decision itself (last workgroup)
    int stage = 0; bool dense_tried = false, handoff = false, window_choice = false;
};
// Longer lines, larger grids (L > 2048, finf + (L + 2)^2 >= 2^(32 - B), 2^31 voxels or more): the marching sweeps with unbounded scans.
DcGeometry envelope_dc_geometry(const sdfgpu_context* h, int stage, int64_t nx, int64_t ny, int64_t nz, int64_t ny_full = -1) {
    DcGeometry g{};
    if (ny_full < 0) ny_full = ny;
    const int64_t L = stage == 2 ? ny : nx;
    if (!h->envelope_dc || L < 1 || L > 2048 || nx * ny * nz >= (1ll << 31)) return g;
    int B = 1;
    while ((1ll << B) < L) ++B;
    const int64_t finf = (nx - 1) * (nx - 1) + (ny_full - 1) * (ny_full - 1) + (nz - 1) * (nz - 1) + 1;   // > every real d^2
    if (finf + (L + 2) * (L + 2) >= (1ll << (32 - B))) return g;
    if ((L + 2) * (2ll << B) >= (1ll << 23)) return g;            // 24-bit multiplier operands
    if (envelope_dc_lds_bytes((int)L) > kDcMaxDynamicLds) return g;
    g.ok = true; g.B = B; g.finf = (uint32_t)finf;
    g.pitch = envelope_dc_pitch((int)L);
    g.M = (int)((L + 7) / 8);
    g.Kp = 0;
    return g;
}

bool far_geometry_ok(const sdfgpu_context* h, int stage, int64_t nx, int64_t ny, int64_t nz, int64_t ny_full = -1) {
    return envelope_dc_geometry(h, stage, nx, ny, nz, ny_full).ok;
}

int launch_decide(sdfgpu_handle h, int stage, bool dense_tried, hipStream_t s, bool handoff, bool window_choice);

// KE2 / KE3: exact far-field sweeps.  guard: run iff (*guard != 0) != guard_invert (nullptr: always).
int launch_envelope(sdfgpu_handle h, int stage, const int16_t* d_in16, const int32_t* d_side_in, void* d_out,
                    int32_t* d_side_out, int64_t nx, int64_t ny, int64_t nz, double resolution, int vb,
                    uint32_t* d_maxdsq, const uint32_t* guard, hipStream_t s, int guard_invert = 0,
                    uint32_t* probe_out = nullptr, const DcExtra* ex = nullptr, const DcDecide* dec = nullptr) {
    (void)d_maxdsq;
    const DcGeometry g = envelope_dc_geometry(h, stage, nx, ny, nz, ex ? ex->ny_glob : -1);
    if (!g.ok) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "shape not taken by the far-field kernel (internal: callers check the geometry first)");
    {
        EnvDcArgs a{};
        a.in16 = d_in16; a.side_in = d_side_in; a.out = d_out; a.side_out = d_side_out;
        int64_t ntiles;
        constexpr int NL = kDcLines;
        if (stage == 2) { a.group_lines = nz; a.tiles_per_outer = (nz + NL - 1) / NL; ntiles = nx * a.tiles_per_outer; a.outer_stride = ny * nz; a.line_stride = nz; a.L = (int)ny; }
        else { a.group_lines = ny * nz; a.tiles_per_outer = (ny * nz + NL - 1) / NL; ntiles = a.tiles_per_outer; a.outer_stride = 0; a.line_stride = ny * nz; a.L = (int)nx; }
        a.B = g.B; a.finf = g.finf; a.pitch = g.pitch; a.M = g.M; a.Kp = 0; a.h = (a.L + 1) / 2;
        a.resolution = resolution; a.vb = vb; a.nx = nx; a.ny = ny; a.nz = nz;
        a.y_off = 0; a.ny_glob = ny;
        if (ex) {
            a.in_i32 = ex->in_i32; a.out_i32 = ex->out_i32; a.y_off = ex->y_off; a.i32_flag = ex->i32_flag;
            if (ex->ny_glob >= 0) a.ny_glob = ex->ny_glob;
            a.ran_flag = ex->ran_flag;
            if (!probe_out) { a.row_bits = ex->row_bits; a.row_words = ex->row_words; a.plane_any = ex->plane_any; a.some_empty = h->d_small + 22; a.flat_on = h->flat_tiles; a.flat_score = h->d_small + 40; }
            if (stage == 2) { a.bits = ex->bits; a.nzw = ex->nzw; }
            if (stage == 3 && ex->loop && ex->fold_ticket && !probe_out) {
                a.fold_status = h->d_small; a.fold_result = ex->fold_result; a.fold_report = ex->fold_report; a.fold_ticket = ex->fold_ticket;
                a.fold_report_mask = kReportDense;
            }
        }
        const bool loop = ex && ex->loop && !probe_out;
        a.maxdsq = h->d_slots; a.guard = guard; a.guard_invert = guard_invert;
#ifdef SDFGPU_DEBUG_HOOKS
        a.dbg = (h->dc_debug_stage == 0 || h->dc_debug_stage == stage) ? h->dc_debug : 0;    // (an ablated y sweep hands garbage to the x sweep: ablate one stage at a time)
#endif
#ifdef SDFGPU_PHASE_CLOCKS
        if (!h->d_clocks) { HIP_TRY(h, hipMalloc((void**)&h->d_clocks, 16 * 8)); HIP_TRY(h, hipMemset(h->d_clocks, 0, 16 * 8)); }
        a.clocks = h->d_clocks;
#endif
        if (ntiles > 0x7fffffffLL) return fail(h, SDFGPU_ERR_UNSUPPORTED_SIZE, "envelope grid too large");
        const int64_t all = ntiles;
        if (probe_out && dec && h->probe_window) {
            // the window statistic (k_probe_window): thresholds up to 81 = windows up to radius 8
            const int thr = h->far_thr[stage - 2], thr2 = stage == 2 ? h->mid_thr_y : 0, thr3 = stage == 2 ? h->far_thr[1] : 0;
            int W = 0;
            while ((W + 1) * (W + 1) < std::max(thr, std::max(thr2, thr3))) ++W;
            const bool handoff_in = ex && ex->i32_flag;                      // p16 shape: int32 values only behind the flag, which ends the probe
            const int32_t* in32 = (stage == 3 && ex && ex->in_i32 && !handoff_in) ? ex->in_i32 : nullptr;
            const int16_t* in16 = in32 ? nullptr : d_in16;
            if (W <= 8 && (in16 || in32)) {
                ProbeArgs pa{};
                pa.in16 = in16; pa.in32 = in32;
                pa.n = nx * ny * nz; pa.ls = a.line_stride; pa.L = a.L; pa.W = W;
                pa.thr = thr; pa.thr2 = thr2; pa.thr3 = thr3;
                pa.nsamples = (uint32_t)std::min<int64_t>(pa.n, 32768);
                pa.step = std::max<int64_t>(1, pa.n / pa.nsamples);
                if (pa.step > 0xffffff) pa.step = 0xffffff;                 // (the kernel hashes the offset inside a stretch in 24 bits)
                pa.probe_out = probe_out; pa.guard = guard; pa.i32_flag = handoff_in ? ex->i32_flag : nullptr;
                pa.decide_small = h->d_small; pa.decide_stage = dec->stage; pa.decide_dense_tried = dec->dense_tried ? 1 : 0;
                pa.decide_force = h->force_env; pa.decide_den = h->far_den[dec->stage]; pa.decide_handoff = dec->handoff ? 1 : 0;
                pa.decide_mid_den = (dec->stage == 0 && dec->window_choice) ? h->mid_den_y : 0;
                pa.decide_xden = h->far_den[1];
                const unsigned nb = (pa.nsamples + 255u) / 256u;
                if (stage == 2 && W <= 3) hipLaunchKernelGGL((k_probe_window<2, 3>), dim3(nb), dim3(256), 0, s, pa);
                else if (stage == 2) hipLaunchKernelGGL((k_probe_window<2, 8>), dim3(nb), dim3(256), 0, s, pa);
                else if (W <= 3) hipLaunchKernelGGL((k_probe_window<3, 3>), dim3(nb), dim3(256), 0, s, pa);
                else hipLaunchKernelGGL((k_probe_window<3, 8>), dim3(nb), dim3(256), 0, s, pa);
                HIP_TRY(h, hipGetLastError());
                return SDFGPU_OK;
            }
        }
        if (probe_out) {                                        // sample ~256 tiles spread over the grid, store nothing
            a.probe_stride = (int)std::max<int64_t>(1, std::min<int64_t>(64, ntiles / 256));
            a.probe_thr = h->far_thr[stage - 2];
            a.probe_thr2 = stage == 2 ? h->mid_thr_y : 0;
            a.probe_thr3 = (stage == 2 && dec) ? h->far_thr[1] : 0;
            a.probe_out = probe_out;
            if (dec) {
                a.decide_small = h->d_small; a.decide_stage = dec->stage; a.decide_dense_tried = dec->dense_tried ? 1 : 0;
                a.decide_force = h->force_env; a.decide_den = h->far_den[dec->stage]; a.decide_handoff = dec->handoff ? 1 : 0;
                a.decide_mid_den = (dec->stage == 0 && dec->window_choice) ? h->mid_den_y : 0;
                a.decide_xden = h->far_den[1];
            }
            ntiles = (ntiles + a.probe_stride - 1) / a.probe_stride;
            while (ntiles > 0 && (ntiles - 1) * a.probe_stride + ((ntiles - 1) * 7) % a.probe_stride >= all) --ntiles;
            if (ntiles == 0) return dec ? launch_decide(h, dec->stage, dec->dense_tried, s, dec->handoff, dec->window_choice) : SDFGPU_OK;
        }
        // vector loads: 4 consecutive lines per load, whole tiles, aligned rows
        auto al = [](const void* p, uintptr_t n) { return (reinterpret_cast<uintptr_t>(p) % n) == 0; };
        const bool vec = !a.bits && (nz % 4) == 0 && (a.group_lines % NL) == 0 && al(d_in16, 8) && al(d_side_in, 16) && al(a.in_i32, 16);
#ifdef SDFGPU_DEBUG_HOOKS
        const size_t lds = envelope_dc_lds_bytes(a.L, NL) + (size_t)(h->dc_debug >> 8) * 1024;  // profiling builds: LDS padding = lower occupancy
#else
        const size_t lds = envelope_dc_lds_bytes(a.L, NL);
#endif
        // (the kernel is a template over lines per tile and lanes; measured at 512^3: 8-line tiles -- 20 KB of LDS, 7 - 8
        //  workgroups per CU -- with 128 or 256 lanes are 12 - 25 % slower than 16 lines x 256 lanes, 16 lines x 512 lanes
        //  +-5 %, 32 lines x 512 lanes +-3 %; 2 instead of 4 workgroups per CU is 1.55x slower)
        // lines above 512: the tile's keys take 66 KB and more, two workgroups per CU -- of 512 lanes then (round 4: with 256 they
        // left the CU at 2 waves per SIMD, and 1024-voxel lines cost 1.5 - 2x their share).  Round 5 tried the other way out, 8 lines
        // x 256 lanes (38 KB: four workgroups per CU again) on the 1024^3 scenes: same fields bit for bit, room 18.1 ms instead of
        // 16.0, boxes / shells / spheres within 2 % -- the long lines are not short of resident workgroups
        const bool big = a.L > 512 && !loop;
        const int NT = big ? 512 : 256;
        a.ntiles = ntiles;
        // LOOP form: at most 2048 workgroups (a guarded exit costs 1.7 us up to there and grows with the grid), a multiple of 8
        // so that a workgroup's tiles stay on its XCD
        const int64_t nwg = loop ? std::min<int64_t>(ntiles, h->standby_grid) : ntiles;
        // 512-voxel lines, vector loads, one tile per workgroup: the instances with the line's geometry as compile-time constants (option "dc_fixed")
        const bool fixed512 = h->dc_fixed && a.L == 512 && vec && !loop && !big && a.B == 9 && a.pitch == 514 && a.M == 64 && a.h == 256;
        const bool fixed1024 = h->dc_fixed && a.L == 1024 && vec && big && a.B == 10 && a.pitch == 1026 && a.M == 128 && a.h == 512;
        const int which = fixed512 ? 12 + (stage == 3 ? 1 : 0) : fixed1024 ? 14 + (stage == 3 ? 1 : 0) :
                          (stage == 3 ? 1 : 0) + (vec ? 2 : 0) + (loop ? 4 : 0) + (big ? 8 : 0);
        auto launch = [&](auto kern) -> int {
            // (the attribute is per kernel: raised once per instantiation, not on every launch -- ADVICE r3)
            if (lds > 64 * 1024 && !h->dc_attr_set[which]) {
                HIP_TRY(h, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDcMaxDynamicLds));
                h->dc_attr_set[which] = true;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3((unsigned)NT), lds, s, a);
            return SDFGPU_OK;
        };
        int rc;
        switch (which) {
            case 0: rc = launch(k_envelope_dc<2, false, 256, 16>); break;
            case 1: rc = launch(k_envelope_dc<3, false, 256, 16>); break;
            case 2: rc = launch(k_envelope_dc<2, true, 256, 16>); break;
            case 3: rc = launch(k_envelope_dc<3, true, 256, 16>); break;
            case 4: rc = launch(k_envelope_dc<2, false, 256, 16, true>); break;
            case 5: rc = launch(k_envelope_dc<3, false, 256, 16, true>); break;
            case 6: rc = launch(k_envelope_dc<2, true, 256, 16, true>); break;
            case 7: rc = launch(k_envelope_dc<3, true, 256, 16, true>); break;
            case 8: rc = launch(k_envelope_dc<2, false, 512, 16, false, 4>); break;
            case 9: rc = launch(k_envelope_dc<3, false, 512, 16, false, 4>); break;
            case 10: rc = launch(k_envelope_dc<2, true, 512, 16, false, 4>); break;
            case 11: rc = launch(k_envelope_dc<3, true, 512, 16, false, 4>); break;
            case 12: rc = launch(k_envelope_dc<2, true, 256, 16, false, 4, 512>); break;
            case 13: rc = launch(k_envelope_dc<3, true, 256, 16, false, 4, 512>); break;
            case 14: rc = launch(k_envelope_dc<2, true, 512, 16, false, 4, 1024>); break;
            default: rc = launch(k_envelope_dc<3, true, 512, 16, false, 4, 1024>); break;
        }
        if (rc) return rc;
        HIP_TRY(h, hipGetLastError());
        return SDFGPU_OK;
    }
}

int launch_decide(sdfgpu_handle h, int stage, bool dense_tried, hipStream_t s, bool handoff = false, bool window_choice = false);
int launch_decide(sdfgpu_handle h, int stage, bool dense_tried, hipStream_t s, bool handoff, bool window_choice) {
    hipLaunchKernelGGL(k_decide_tier, dim3(1), dim3(1), 0, s, h->d_small, stage, dense_tried ? 1 : 0, h->force_env,
                       1, h->far_den[stage], handoff ? 1 : 0, (stage == 0 && window_choice) ? h->mid_den_y : 0);
    HIP_TRY(h, hipGetLastError());
    return SDFGPU_OK;
}

// K0 + KD: pack to bits, then the bit-parallel ball kernel (sdfgpu_dense.hpp)
bool dense_eligible(const sdfgpu_context* h, int64_t nz, int vb) {
    const int64_t nzw = nz / 32;
    (void)vb;                                     // (the tuned ball kernel folds the virtual border itself)
    return h->dense_on && (nz % 32) == 0 && nzw >= 1 && nzw <= 64 && (nzw & (nzw - 1)) == 0;
}

int launch_pack_bits(sdfgpu_handle h, const uint8_t* d_mask, const void* d_cells, size_t stride, size_t off, int unknown,
                     int64_t n, uint32_t* d_bits, hipStream_t s) {
    if (d_cells) {
        CellLoader ld{reinterpret_cast<const char*>(d_cells), (int64_t)stride, (int64_t)off, unknown};
        hipLaunchKernelGGL(k_pack_bits_generic<CellLoader>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                           ld, d_bits, n);
    } else if ((reinterpret_cast<uintptr_t>(d_mask) % 16) == 0) {
        const int64_t n16 = n / 16;
        auto blocks = [&](int chunks) { const int64_t pb = (int64_t)kBlock * chunks; return dim3((unsigned)((n16 + pb - 1) / pb)); };
        switch (h->pack_variant) {
            case 1: hipLaunchKernelGGL((k_pack_bits_mask<1, false>), blocks(1), dim3(kBlock), 0, s, d_mask, d_bits, n16); break;
            case 2: hipLaunchKernelGGL((k_pack_bits_mask<2, false>), blocks(2), dim3(kBlock), 0, s, d_mask, d_bits, n16); break;
            case 3: hipLaunchKernelGGL((k_pack_bits_mask<8, false>), blocks(8), dim3(kBlock), 0, s, d_mask, d_bits, n16); break;
            case 4: hipLaunchKernelGGL((k_pack_bits_mask<4, true>), blocks(4), dim3(kBlock), 0, s, d_mask, d_bits, n16); break;
            case 5: hipLaunchKernelGGL((k_pack_bits_mask<1, true>), blocks(1), dim3(kBlock), 0, s, d_mask, d_bits, n16); break;
            case 6: hipLaunchKernelGGL((k_pack_bits_mask<4, false>), blocks(4), dim3(kBlock), 0, s, d_mask, d_bits, n16); break;
            default: hipLaunchKernelGGL((k_pack_bits_mask<4, true>), blocks(4), dim3(kBlock), 0, s, d_mask, d_bits, n16); break;
        }
    } else {
        MaskLoader ld{d_mask};
        hipLaunchKernelGGL(k_pack_bits_generic<MaskLoader>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                           ld, d_bits, n);
    }
    HIP_TRY(h, hipGetLastError());
    return SDFGPU_OK;
}

// Folds the per-slot maxima of the final-stage kernels launched so far into d_maxdsq[0..1] and clears the slots.
int fold_slots(sdfgpu_handle h, uint32_t* d_maxdsq, hipStream_t s, uint32_t* result = nullptr, uint32_t* report = nullptr,
               uint32_t report_mask = kReportDense) {
    hipLaunchKernelGGL(k_fold_slots, dim3(1), dim3(kSlots), 0, s, h->d_slots, d_maxdsq, result, report, report_mask);
    HIP_TRY(h, hipGetLastError());
    return SDFGPU_OK;
}

int launch_ball_dense(sdfgpu_handle h, const uint32_t* d_bits, float* d_out, int64_t rows_x, int64_t out_lo, int64_t out_hi,
                      int64_t ny, int64_t nz, double resolution, uint32_t* d_maxdsq, uint32_t* d_uncert, hipStream_t s,
                      uint32_t* d_fix_needed = nullptr, bool early_out = false, int vb = 0, int64_t nx_glob = 0, int radius = 2,
                      const uint32_t* d_guard = nullptr) {
    const int R = radius == 3 ? kBall3R : kBallR;                   // 3: KD3 (sdfgpu_dense3.hpp; whole-grid builds without virtual border only)
    DenseArgs a{};
    a.early_out = early_out ? 1 : 0;
    a.vb = vb; a.nx_glob = (int)nx_glob;
    a.bits = d_bits; a.out = d_out;
    a.nzw = (int)(nz / 32);
    a.log2_nzw = 0;
    while ((1 << a.log2_nzw) < a.nzw) ++a.log2_nzw;
    a.ny = (int)ny; a.rows_x = (int)rows_x; a.out_lo = (int)out_lo; a.out_hi = (int)out_hi;
    int bd = h->ball_block > 0 ? h->ball_block : 256;
    if (bd != 256 && bd != 512 && bd != 1024) bd = 256;
    while (bd / a.nzw < 1) bd *= 2;
    const int rows = bd / a.nzw;                           // tile rows per workgroup
    int best_tx = 1, best_ty = rows, best_cost = 1 << 30;
    for (int ty = 1; ty <= rows; ty *= 2) {                // smallest halo-inclusive footprint
        const int tx = rows / ty;
        const int cost = (tx + 2 * R) * (ty + 2 * R);
        if (cost < best_cost || (cost == best_cost && ty > best_ty)) { best_cost = cost; best_tx = tx; best_ty = ty; }
    }
    a.tx = best_tx; a.ty = best_ty;
    a.log2_ty = 0;
    while ((1 << a.log2_ty) < a.ty) ++a.log2_ty;
    a.inv_hy = (65536 + (a.ty + 2 * R) - 1) / (a.ty + 2 * R);
    static const int level_d2[7] = {1, 2, 3, 4, 5, 6, 8};
    for (int l = 0; l < 7; ++l) a.mag[l] = (float)(std::sqrt((double)level_d2[l]) * resolution);
    a.mag[7] = 0.0f;
    static const int level3_d2[13] = {1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14};
    for (int l = 0; l < 16; ++l) a.mag3[l] = l < 13 ? (float)(std::sqrt((double)level3_d2[l]) * resolution) : 0.0f;
    a.slots = h->d_slots; a.uncertified = d_uncert;
    a.reason = early_out ? h->d_small + 21 : nullptr;                // (whole builds: the context's status block; stage calls have none)
    a.nt_store = h->nt_store;
    a.guard = d_guard;
    // KD3 gives up at once when a wave holds more undecided voxels than the stage behind it takes: KF's per-tile cap -- or, with
    // the shell pass in between (which turns words, not voxels, around and leaves KF what lies beyond d^2 = 36), three quarters
    // of the wave: the faces of a sparse noise grid, where a voxel sees a fraction of the ball, used to end the tier there
    const bool shell = radius == 3 && h->shell_on && d_fix_needed && early_out;      // (whole builds: the status block carries the sample)
    a.max_undecided = shell ? 1536 : kBall3MaxUndecided;
    a.und_sample = shell ? h->d_slots : nullptr;
    const int64_t gx = (ny + a.ty - 1) / a.ty, gy = (out_hi - out_lo + a.tx - 1) / a.tx;
    if (gx > 0x7fffffffLL || gy > 65535) return fail(h, SDFGPU_ERR_UNSUPPORTED_SIZE, "dense grid too large");
    const size_t pitch = a.nzw < 32 ? a.nzw + 32 : a.nzw + 2;        // must match k_ball_dense
    const size_t tile_words = ((size_t)(a.tx + 2 * R) * (a.ty + 2 * R) * pitch + 3) & ~(size_t)3;
    const size_t lds = radius == 3 ? tile_words * 4 + (size_t)bd * 20 + 1024 * 8 + 128 : tile_words * 4 + (size_t)bd * 16 + 256 * 8 + 64;
    const dim3 grid((unsigned)gx, (unsigned)gy);
    if (d_fix_needed) {                                              // fix-up mode: hand the undecided voxels to KF
        const size_t tiles = (size_t)gx * gy;
        // (the undecided bits of a STAGED fix-up stage live in the z field's storage, which the general pipeline only writes
        //  after KF has consumed them: a fresh context's first build does not pay a 17 MB allocation for a stage that a
        //  far-field scene leaves at once)
        if (!h->unc_override) if (int rc = ensure(h, h->unc, (size_t)(out_hi - out_lo) * ny * a.nzw * 4, "undecided words")) return rc;
        uint32_t* tileflag = nullptr;
        if (h->unc_override && h->unc_override_bytes >= (size_t)(out_hi - out_lo) * ny * a.nzw * 4 + 256 + tiles * 4) {
            // ... and so do its tile flags (behind the bits; cleared here: the storage is not ours between builds)
            tileflag = h->unc_override + ((size_t)(out_hi - out_lo) * ny * a.nzw + 63) / 64 * 64;
            HIP_TRY(h, hipMemsetAsync(tileflag, 0, tiles * 4, s));
        } else {
            if (h->tileflag.bytes < tiles * 4 || !h->tileflag.ptr) {
                if (int rc = ensure(h, h->tileflag, tiles * 4, "tile flags")) return rc;
                HIP_TRY(h, hipMemsetAsync(h->tileflag.ptr, 0, h->tileflag.bytes, s));     // afterwards KF keeps it zero
            }
            tileflag = (uint32_t*)h->tileflag.ptr;
        }
        a.unc = h->unc_override ? h->unc_override : (uint32_t*)h->unc.ptr; a.tileflag = tileflag; a.fix_needed = d_fix_needed;
    }
    const bool zinv = nz <= (int64_t)bd * 4 && !(h->ball_variant & 2);   // an expansion pass covers whole z-rows
    a.checked = h->ball_variant & 1;
    if (radius == 3) {
        if (bd != 256 || vb) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "KD3: 256-lane tiles without virtual border only (internal)");
        // nz = 512 (16 words per row, 4 x 4-row tiles): the instance with compile-time row pitch and halo width (option "dense3_fixed")
        if (zinv && a.nzw == 16 && a.ty == 4 && a.tx == 4 && h->dense3_fixed) hipLaunchKernelGGL((k_ball_dense3<256, true, 16, 4>), grid, dim3(256), lds, s, a);
        else if (zinv) hipLaunchKernelGGL((k_ball_dense3<256, true>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((k_ball_dense3<256, false>), grid, dim3(256), lds, s, a);
    }
    else if (bd == 1024) hipLaunchKernelGGL((k_ball_dense<1024, true>), grid, dim3(1024), lds, s, a);
    else if (bd == 512) hipLaunchKernelGGL((k_ball_dense<512, true>), grid, dim3(512), lds, s, a);
    else if (zinv) hipLaunchKernelGGL((k_ball_dense<256, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((k_ball_dense<256, false>), grid, dim3(256), lds, s, a);
    if (d_fix_needed) {
        HIP_TRY(h, hipGetLastError());
        if (!h->fix_order.ptr) {                                     // (dx, dy) rows by increasing dx^2 + dy^2, built once
            std::vector<uint32_t> order;
            for (int dx = -kFixR; dx <= kFixR; ++dx)
                for (int dy = -kFixR; dy <= kFixR; ++dy)
                    order.push_back((uint32_t)(dx + kFixR) | ((uint32_t)(dy + kFixR) << 8) | ((uint32_t)(dx * dx + dy * dy) << 16));
            std::stable_sort(order.begin(), order.end(), [](uint32_t x, uint32_t y) { return (x >> 16) < (y >> 16); });
            if (int rc = ensure(h, h->fix_order, order.size() * 4, "fix-up row order")) return rc;
            HIP_TRY(h, hipMemcpy(h->fix_order.ptr, order.data(), order.size() * 4, hipMemcpyHostToDevice));
        }
        // KD6 (round 5): the bit-parallel shell pass 16 <= d^2 <= 36 over the words KD3 left undecided voxels in, in front of KF
        // (which then only sees what lies beyond d^2 = 36) -- sdfgpu_dense6.hpp.  Behind KD3 only: KD's undecided voxels start
        // at d^2 = 9, below the shell.
        if (shell && bd == 256) {
            ShellArgs sa{};
            sa.bits = d_bits; sa.out = d_out; sa.unc = a.unc; sa.tileflag = a.tileflag; sa.fix_needed = d_fix_needed;
            sa.uncertified = d_uncert;
            // budget: 1 / shell_budget_den of the voxels undecided behind KD3 (the sample holds 1 / 16 of them; small grids: at least
            // one wave's worth)
            const uint32_t budget = (uint32_t)std::max<int64_t>((out_hi - out_lo) * ny * nz / h->shell_budget_den / 16, 2048);
            hipLaunchKernelGGL(k_shell_budget, dim3(1), dim3(kSlots), 0, s, h->d_slots, d_fix_needed, d_uncert, a.reason, budget);
            sa.nzw = a.nzw; sa.log2_nzw = a.log2_nzw; sa.ny = a.ny; sa.rows_x = a.rows_x; sa.out_lo = a.out_lo; sa.out_hi = a.out_hi;
            sa.tx = a.tx; sa.ty = a.ty; sa.log2_ty = a.log2_ty; sa.resolution = resolution; sa.slots = h->d_slots; sa.min_words = h->shell_min_words;
            const size_t slds = shell_lds_bytes(256, a.tx, a.ty, a.nzw);
            if (slds <= 64 * 1024) {
                const dim3 sgrid((unsigned)((gx + kShellGroup - 1) / kShellGroup), (unsigned)gy);       // kShellGroup tiles (along y) per workgroup
                hipLaunchKernelGGL(k_ball_shell<256>, sgrid, dim3(256), slds, s, sa);
                HIP_TRY(h, hipGetLastError());
            }
        }
        FixArgs f{};
        f.bits = d_bits; f.out = d_out; f.unc = a.unc; f.tileflag = a.tileflag; f.fix_needed = d_fix_needed;
        f.order = (const uint32_t*)h->fix_order.ptr;
        f.nzw = a.nzw; f.log2_nzw = a.log2_nzw; f.ny = a.ny; f.rows_x = a.rows_x; f.out_lo = a.out_lo; f.out_hi = a.out_hi;
        f.tx = a.tx; f.ty = a.ty; f.log2_ty = a.log2_ty; f.resolution = resolution;
        f.slots = h->d_slots; f.uncertified = d_uncert; f.reason = a.reason;
        const size_t flds = (size_t)(2 * kFixOrderPad + kFixCap + 4) * 4 +
                            (size_t)(a.tx + 2 * kFixTileR) * (a.ty + 2 * kFixTileR) * (a.nzw + 2) * 4;
        if (bd == 1024) hipLaunchKernelGGL(k_ball_fixup<1024>, grid, dim3(1024), flds, s, f);
        else if (bd == 512) hipLaunchKernelGGL(k_ball_fixup<512>, grid, dim3(512), flds, s, f);
        else hipLaunchKernelGGL(k_ball_fixup<256>, grid, dim3(256), flds, s, f);
    }
    HIP_TRY(h, hipGetLastError());
    return SDFGPU_OK;
}

// Generic dense tier (any nz, virtual border): k_pack_bits_rows + k_ball_dense_generic
int launch_dense_generic(sdfgpu_handle h, const uint8_t* d_mask, const void* d_cells, size_t stride, size_t off, int unknown,
                         int64_t nx, int64_t ny, int64_t nz, double resolution, int vb, float* d_out, uint32_t* d_uncert,
                         hipStream_t s, const uint32_t* d_bits_in = nullptr) {
    const int64_t nzw = (nz + 31) / 32, nrows = nx * ny, nwords = nrows * nzw;
    if (nwords > 0x7fffffffLL * (int64_t)kBlock) return fail(h, SDFGPU_ERR_UNSUPPORTED_SIZE, "dense grid too large");
    if (int rc = ensure(h, h->bits, (size_t)nwords * 4, "bit field")) return rc;
    const dim3 grid((unsigned)((nwords + kBlock - 1) / kBlock)), block(kBlock);
    if (d_bits_in) {                                            // (linear bits -> rows padded to whole words)
        BitsLoader ld{d_bits_in};
        hipLaunchKernelGGL(k_pack_bits_rows<BitsLoader>, grid, block, 0, s, ld, (uint32_t*)h->bits.ptr, nrows, (int)nz, (int)nzw);
    } else if (d_cells) {
        CellLoader ld{reinterpret_cast<const char*>(d_cells), (int64_t)stride, (int64_t)off, unknown};
        hipLaunchKernelGGL(k_pack_bits_rows<CellLoader>, grid, block, 0, s, ld, (uint32_t*)h->bits.ptr, nrows, (int)nz, (int)nzw);
    } else {
        MaskLoader ld{d_mask};
        hipLaunchKernelGGL(k_pack_bits_rows<MaskLoader>, grid, block, 0, s, ld, (uint32_t*)h->bits.ptr, nrows, (int)nz, (int)nzw);
    }
    HIP_TRY(h, hipGetLastError());
    DenseGenArgs a{};
    a.bits = (const uint32_t*)h->bits.ptr; a.out = d_out;
    a.nx = (int)nx; a.ny = (int)ny; a.nz = (int)nz; a.nzw = (int)nzw; a.vb = vb;
    static const int level_d2[7] = {1, 2, 3, 4, 5, 6, 8};
    for (int l = 0; l < 7; ++l) a.mag[l] = (float)(std::sqrt((double)level_d2[l]) * resolution);
    a.mag[7] = 0.0f;
    a.slots = h->d_slots; a.uncertified = d_uncert;
    hipLaunchKernelGGL(k_ball_dense_generic, grid, block, 0, s, a);
    HIP_TRY(h, hipGetLastError());
    return SDFGPU_OK;
}

// d_bits_in != nullptr (round 6): the occupancy arrives as one bit per voxel, linear order (sdfgpu_build_bits*): the tuned dense
// kernels read it in place (no K0), the z sweep and the generic dense tier read it through their bit loaders -- no byte mask exists.
int build_device_impl(sdfgpu_handle h, const uint8_t* d_filled, const void* d_cells, size_t stride, size_t off,
                      int unknown, int64_t nx, int64_t ny, int64_t nz, double resolution, int vb,
                      float* d_out, hipStream_t s, const uint32_t* d_bits_in = nullptr) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if ((!d_filled && !d_cells && !d_bits_in) || !d_out) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null device pointer");
    if (d_bits_in && (reinterpret_cast<uintptr_t>(d_bits_in) % 4) != 0) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "the bit field must be 4-byte aligned");
    if (int rc = check_dims(h, nx, ny, nz)) return rc;
    if (d_cells && (stride < 4 || (stride % 4) || (off % 4) || off + 4 > stride))
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "cell_stride/occupancy_offset must be 4-byte aligned and in range");
    HIP_TRY(h, hipSetDevice(h->device));
    canonical_dims(nx, ny, nz);
    const int64_t n = nx * ny * nz;
    const bool p16 = plane16_eligible(h, ny, nz);
    if (int rc = ensure(h, h->yzfield, (size_t)n * 4, "int32 plane field / side table")) return rc;          // int32 plane field / side table
    if (p16) if (int rc = ensure(h, h->plane16, (size_t)n * 2, "16-bit plane field")) return rc;
    void* zy_out = p16 ? h->plane16.ptr : h->yzfield.ptr;
    int32_t* zy_side = p16 ? (int32_t*)h->yzfield.ptr : nullptr;
    bool dense = dense_eligible(h, nz, vb) && ny <= 0x7fffffff;
    // shapes / modes the tuned dense kernels do not take go through their generic forms (any nz, virtual border)
    const bool dense_generic = !dense && h->dense_on && h->dense_generic_on && nx <= 0x7fffffff && ny <= 0x7fffffff;
    dense = dense || dense_generic;
    // Learn from an earlier build on this handle, if its status block has arrived (never a wait), and plan the dense tier of
    // this one: sdfgpu_policy.hpp.  Every outcome is exact; the policy only decides whether and in which form the DENSE tier is
    // enqueued and whether the pipeline behind it may be the two-launch stand-by (which tier does a sweep is decided on the
    // device inside the build).
    if (h->flags_pending && hipEventQuery(h->flags_ev) == hipSuccess) {
        h->flags_pending = false;
        //   (a staged build -- KD, then KD3 + KF on KD's verdict -- reports KD's own verdict, status word 20, as word 8)
        h->pol.consume_report(h->h_flags[3] != 0, h->h_flags[6] != 0, h->h_flags[8] != 0);
        // (a predicted build's flags are raised by the launches themselves: they teach the habit nothing, and a near-field scene would
        //  keep a stale "far" alive through them -- ADVICE r5)
        if (!h->pol.prev.predicted) h->far.consume_report(h->h_flags[4] != 0, h->h_flags[5] != 0);
    }
    if (h->far_pending && hipEventQuery(h->far_ev) == hipSuccess) {
        h->far_pending = false;
        h->far.consume_report(h->h_far[4] != 0, h->h_far[5] != 0);
    }
    const DensePlan plan = h->pol.plan(dense, dense_generic, nz / 32 <= 256 && h->ball_block <= 256, vb != 0);
    dense = plan.dense;
    // bounded marching scans + the far-field kernel behind them, on every shape that kernel takes; other shapes (lines beyond
    // 2048, keys beyond 32 bits) keep unbounded marching scans.  Shapes without the 16-bit plane field (nz % 4 != 0) hand
    // exact int32 plane values from the y to the x sweep.
    const bool far_ok = h->envelope_on && far_geometry_ok(h, 2, nx, ny, nz) && far_geometry_ok(h, 3, nx, ny, nz);
    const bool envelope = far_ok && !(h->pol.expect_dense && dense);
    // Stand-by behind a TRUSTED dense tier (round 4).  It nearly always exits on its guard, so it must be
    // few launches with small grids -- and since the scene of a stream can change under the handle (dense -> far-field), it
    // must be BOUNDED whatever the scene turns into.  That is the far-field pair with no probes and no marching sweeps:
    // KE2 [guard] -> KE3 [guard], int32 hand-off, LOOP form (1024 workgroups each); KE2 takes its z distances straight from
    // the dense tier's bit field (zdist_from_bits), so no z sweep is launched in front: TWO guarded launches like the
    // K12 + K3/16 pair it replaces (a third one, a guarded K1, cost the dense-certified step 2.6 us: 0.1405 -> 0.1431 ms),
    // O(L log L) per line on any input (the old pair ran unbounded outward scans: tens of ms on the build in which a
    // dense scene turned into the two-box cloud).
    const bool standby = far_ok && h->standby_far && h->pol.expect_dense && dense && !h->fused_always;
    // Device-side tier selection: the marching-vs-far-field choice of each axis is made INSIDE this build from a probe of the
    // sweep's own input, so a fresh context (the reference's API is one-shot: collision_map.hpp:680-712 builds and returns)
    // never runs a sweep that is thrown away, and a handle's latency does not depend on what it built before.
    const bool dev_select = envelope;
    // K12 (fused z+y) only as the guarded stand-by behind a dense build that is expected to be certified
    // again; everywhere else K1 + K2 (rows from the int16 z field) scan much faster than the fused kernel
    // recomputes, and only they can hand a far-field y sweep to the envelope kernel.
    const bool fused = !standby && !d_cells && fused_zy_eligible(h, d_filled, zy_out, nz) &&
                       (h->fused_always || (dense && h->pol.expect_dense) || (!envelope && !dense));
    const bool select = dev_select && !fused;
    // (see far_predict: the handle's recent builds were far-field on both axes -- or the caller forces it)
    const bool predicted = h->far.plan(select, h->force_env >= 0);
    if (!fused && !standby) if (int rc = ensure(h, h->zfield, (size_t)n * 2, "z field")) return rc;
    // status block [0..7]: maxima, status, uncertified, far flags, fix_needed.  Normally still zero from the previous
    // build's fold kernel; cleared here after a build that failed half-way (or before the first one)
    // Builds on one handle share its status block, extrema slots and scratch fields.  On the same stream they are
    // ordered by the stream; a build issued on ANOTHER stream first waits for the previous build's last kernel.
    if (h->order_valid && s != h->order_stream) HIP_TRY(h, hipStreamWaitEvent(s, h->build_done_ev, 0));
    if (!h->small_clean) {
        HIP_TRY(h, hipMemsetAsync(h->d_small, 0, 128, s));
        HIP_TRY(h, hipMemsetAsync(h->d_slots, 0, (size_t)kSlots * kSlotWords * 4, s));   // a failed build may have left maxima behind
    }
    h->small_clean = false;
    // profiling marks: an event is recorded only behind a stage that launched something; a stage that
    // was not launched shares the previous mark (elapsed 0), so profiling adds as few packets as possible
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool launched_since_mark = true;
    // level 3 = level 2 on every 4th build (a sampled timing costs the loop a quarter of the event overhead)
    int prof = h->profiling;
    if (prof == 3) prof = ((h->profiled_builds++ & 3u) == 0 && dense) ? 2 : 0;
    auto mark = [&](int k) -> hipError_t {
        if (!prof) return hipSuccess;
        if (prof == 2 && dense) {
            // only the dominant kernel of the dense path is bracketed: stage 1 = [ev1, ev2], every other stage
            // shares a mark with its neighbour (elapsed 0)
            if (k == 0) return hipSuccess;
            if (k > 2) { ev[k] = ev[k - 1]; return hipSuccess; }
        } else if (k > 0 && !launched_since_mark) {
            ev[k] = ev[k - 1];
            return hipSuccess;
        }
        if (h->event_pool.empty()) {
            hipError_t e = hipEventCreate(&ev[k]);
            if (e != hipSuccess) return e;
        } else {
            ev[k] = h->event_pool.back();
            h->event_pool.pop_back();
        }
        if (k == 1 && ev[0] == nullptr) ev[0] = ev[1];
        launched_since_mark = false;
        return hipEventRecord(ev[k], s);
    };
    HIP_TRY(h, mark(0));
    // Dense path first: exact wherever the nearest opposite voxel is within d^2 <= 8; raises
    // d_small[3] otherwise, in which case (and only then) the general pipeline below does any work.
    bool cur_fix_mode = false, cur_dense3 = false, cur_staged = false;
    const uint32_t* dense_bits = nullptr;                       // the bit field the tuned dense kernels (and the stand-by y sweep) read
    uint8_t* row_any = nullptr;                                 // plane sparsity: one byte per z row, one per x-plane behind them (predicted far-field builds)
    h->last_dense = dense;
    h->guard = nullptr;
    if (dense && dense_generic) {
        HIP_TRY(h, mark(1));
        if (int rc = launch_dense_generic(h, d_filled, d_cells, stride, off, unknown, nx, ny, nz, resolution, vb, d_out,
                                          h->d_small + 3, s, d_bits_in)) return rc;
        launched_since_mark = true;
        h->guard = h->d_small + 3;
    } else if (dense) {
        if (d_bits_in && (reinterpret_cast<uintptr_t>(d_bits_in) % 16) == 0) {
            dense_bits = d_bits_in;                             // (nz % 32 == 0: the caller's linear bit field IS the [x][y][nz / 32] field K0 writes)
        } else if (d_bits_in) {                                 // (the dense kernels stage bit rows with 16-byte loads)
            if (int rc = ensure(h, h->bits, (size_t)n / 8, "bit field")) return rc;
            HIP_TRY(h, hipMemcpyAsync(h->bits.ptr, d_bits_in, (size_t)n / 8, hipMemcpyDeviceToDevice, s));
            dense_bits = (const uint32_t*)h->bits.ptr;
        } else {
            if (int rc = ensure(h, h->bits, (size_t)n / 8, "bit field")) return rc;
            if (int rc = launch_pack_bits(h, d_filled, d_cells, stride, off, unknown, n, (uint32_t*)h->bits.ptr, s)) return rc;
            launched_since_mark = true;
            dense_bits = (const uint32_t*)h->bits.ptr;
        }
        HIP_TRY(h, mark(1));
        // fix-up mode (policy): undecided voxels go to the fix-up kernel, which raises `uncertified` only for what it
        // cannot decide either; otherwise the ball kernel raises it directly and nothing extra is launched
        // (with a virtual border the fix-up kernel stays out: a voxel it would finish may still be bound by b >= 3)
        // (fix-up mode, KD3 in KD's place, the fix-up stage staged behind KD: DensePolicy::plan)
        const bool fix = plan.fix;
        cur_dense3 = plan.dense3;
        cur_staged = plan.staged;
        if (int rc = launch_ball_dense(h, dense_bits, d_out, nx, 0, nx, ny, nz, resolution, h->d_small,
                                       cur_staged ? h->d_small + 20 : h->d_small + 3, s,
                                       (fix || cur_dense3) ? h->d_small + 6 : nullptr, true, vb, nx, cur_dense3 ? 3 : 2)) return rc;
        if (cur_staged) {
            h->unc_override = (!fused && h->zfield.ptr && h->zfield.bytes >= (size_t)n / 8) ? (uint32_t*)h->zfield.ptr : nullptr;
            h->unc_override_bytes = h->unc_override ? h->zfield.bytes : 0;
            const int rc = launch_ball_dense(h, dense_bits, d_out, nx, 0, nx, ny, nz, resolution, h->d_small,
                                             h->d_small + 3, s, h->d_small + 6, true, 0, nx, 3, h->d_small + 20);
            h->unc_override = nullptr;
            if (rc) return rc;
        }
        launched_since_mark = true;
        h->guard = h->d_small + 3;
        cur_fix_mode = fix || cur_dense3;
    } else {
        HIP_TRY(h, mark(1));
    }
    h->last_dense3 = cur_dense3;
    h->last_staged = cur_staged;
    HIP_TRY(h, mark(2));
    h->last_fused = fused;
    h->last_standby = standby;
    h->last_predicted = predicted;
    h->last_plane16 = p16;
    h->far_y = (envelope && !fused) ? h->d_small + 4 : nullptr;
    h->scan_y = h->scan_x = kScanExpectNear;
    if (!fused && !standby) {                                   // (the stand-by pair takes its z distances from the bit field)
        // a build that goes straight to the far-field pair lets the z sweep mark the x-planes that hold a filled voxel at all: a sensed
        // scene leaves most planes without one, and the pair skips them (y sweep: the planes' tiles; x sweep: their row loads)
        if (predicted && h->plane_skip && nx >= 8 && ny <= 1024 && z_wave16_shape(h, nz) && !d_cells &&
            (d_bits_in || (reinterpret_cast<uintptr_t>(d_filled) % 16) == 0)) {
            const size_t o_bits = ((size_t)(nx * ny + nx) + 255) & ~(size_t)255;
            if (int rc = ensure(h, h->planebits, o_bits + (size_t)nx * ((ny + 31) / 32) * 4, "row / x-plane occupancy flags")) return rc;
            row_any = (uint8_t*)h->planebits.ptr;
        }
        if (int rc = launch_sweep_z(h, d_filled, d_cells, stride, off, unknown, nx, ny, nz,
                                    (int16_t*)h->zfield.ptr, s, d_bits_in, row_any)) return rc;
        launched_since_mark = true;
    }
    h->last_plane_skip = row_any != nullptr;
    h->last_dims[0] = nx; h->last_dims[1] = ny; h->last_dims[2] = nz;
    HIP_TRY(h, mark(3));
    // y sweep: marching (bounded scan, may raise far_y) + guarded envelope, or the envelope kernel alone
    const uint32_t* const general_guard = h->guard;           // nullptr, or "the dense tier left voxels undecided"
    // Both axes far-field (decided by the y probe): the y sweep hands the x sweep exact int32 values in the side-table
    // buffer (used whole) instead of p16 + side table -- one store / one load per voxel, no saturation handling, and the
    // x probe is skipped.  d_small[7] carries the choice on the device.
    const bool handoff = select && p16 && h->i32_handoff;        // (shapes without the 16-bit plane field hand int32 values over anyway)
    DcExtra hand2{}, hand3{};
    hand2.out_i32 = (int32_t*)h->yzfield.ptr; hand2.i32_flag = h->d_small + 7;
    hand3.in_i32 = (const int32_t*)h->yzfield.ptr; hand3.i32_flag = h->d_small + 7;
    DcExtra plain2{}, plain3{};                                 // shapes without the 16-bit plane field: int32 in / out, unconditionally
    plain2.out_i32 = (int32_t*)h->yzfield.ptr;
    plain3.in_i32 = (const int32_t*)h->yzfield.ptr;
    const DcExtra* const ex2 = !p16 ? &plain2 : handoff ? &hand2 : nullptr;
    const DcExtra* const ex3 = !p16 ? &plain3 : handoff ? &hand3 : nullptr;
    // (the radius-8 y window exists for 4-voxel lanes only; a forced window leaves nothing to choose)
    const bool window_choice = select && (nz % 4) == 0 && h->march_h != 8 && h->mid_den_y > 0;
    auto decide = [&](int stage) -> int { return launch_decide(h, stage, dense, s, handoff, window_choice); };   // probe counters -> guard words
    DcExtra sb2{}, sb3{};                                       // stand-by pair: int32 hand-off unconditionally, LOOP form, self-reporting
    sb2.out_i32 = (int32_t*)h->yzfield.ptr; sb2.loop = true; sb2.ran_flag = h->d_small + 4;
    sb2.bits = dense_bits ? dense_bits : (const uint32_t*)h->bits.ptr; sb2.nzw = (int)((nz + 31) / 32);
    sb3.in_i32 = (const int32_t*)h->yzfield.ptr; sb3.loop = true; sb3.ran_flag = h->d_small + 5;
    // (moved up: the stand-by x sweep does the fold itself and needs to know where the report goes)
    const bool report = h->envelope_on && h->h_flags_dev && !h->flags_pending && dense;
    bool folded = false;
    if (standby) {
        if (h->standby_fold) {
            sb3.fold_result = h->d_result; sb3.fold_report = report ? h->h_flags_dev : nullptr; sb3.fold_ticket = h->d_small + 23;
            folded = true;
        }
        HIP_TRY(h, mark(4));                                    // (stage slot 3, the marching y sweep: nothing launched)
        if (int rc = launch_envelope(h, 2, nullptr, nullptr, nullptr, nullptr, nx, ny, nz, resolution, vb,
                                     h->d_small, general_guard, s, 0, nullptr, &sb2)) return rc;
        launched_since_mark = true;
        HIP_TRY(h, mark(5));
        HIP_TRY(h, mark(6));                                    // (stage slot 5, the marching x sweep: nothing launched)
        if (int rc = launch_envelope(h, 3, nullptr, nullptr, d_out, nullptr, nx, ny, nz, resolution, vb, h->d_small,
                                     general_guard, s, 0, nullptr, &sb3)) return rc;
        launched_since_mark = true;
    } else if (predicted) {
        // KE2 [general guard] -> KE3 [general guard], exact int32 plane values in between, each launch raising its far flag itself
        DcExtra pr2 = plain2, pr3 = plain3;
        pr2.ran_flag = h->d_small + 4;
        pr3.ran_flag = h->d_small + 5;
        if (row_any) {
            // row bytes -> one bit per row, one byte per x-plane, "some plane is empty" (status word 22)
            uint8_t* plane_any = row_any + nx * ny;
            uint32_t* row_bits = (uint32_t*)(row_any + (((size_t)(nx * ny + nx) + 255) & ~(size_t)255));
            const int row_words = (int)((ny + 31) / 32);
            if (h->flat_score_reset) { h->flat_score_reset = false; HIP_TRY(h, hipMemsetAsync(h->d_small + 40, 0, 8, s)); }
            hipLaunchKernelGGL(k_pack_row_flags, dim3((unsigned)std::min<int64_t>((nx + 15) / 16, 16)), dim3(1024), 0, s, row_any, (int)nx, (int)ny, row_words, row_bits, plane_any, h->d_small + 22, h->d_small + 40);
            HIP_TRY(h, hipGetLastError());
            pr2.row_bits = row_bits; pr2.row_words = row_words; pr2.plane_any = pr3.plane_any = plane_any;
        }
        HIP_TRY(h, mark(4));                                    // (stage slot 3, the marching y sweep: nothing launched)
        if (int rc = launch_envelope(h, 2, (const int16_t*)h->zfield.ptr, nullptr, nullptr, nullptr, nx, ny, nz, resolution, vb,
                                     h->d_small, general_guard, s, 0, nullptr, &pr2)) return rc;
        launched_since_mark = true;
        HIP_TRY(h, mark(5));
        HIP_TRY(h, mark(6));                                    // (stage slot 5, the marching x sweep: nothing launched)
        if (int rc = launch_envelope(h, 3, nullptr, nullptr, d_out, nullptr, nx, ny, nz, resolution, vb, h->d_small,
                                     general_guard, s, 0, nullptr, &pr3)) return rc;
        launched_since_mark = true;
    } else {
    if (select) {
        // probe + decision in one launch (the probe's last workgroup decides); a forced tier needs no probe
        DcDecide dy; dy.stage = 0; dy.dense_tried = dense; dy.handoff = handoff; dy.window_choice = window_choice;
        if (h->force_env < 0) {
            if (int rc = launch_envelope(h, 2, (const int16_t*)h->zfield.ptr, nullptr, h->plane16.ptr, (int32_t*)h->yzfield.ptr,
                                         nx, ny, nz, resolution, vb, h->d_small, general_guard, s, 0, h->d_small + 12,
                                         !p16 ? &plain2 : nullptr, &dy)) return rc;
        } else if (int rc = decide(0)) return rc;
        // marching y sweep: general pipeline needed AND near-field; the probe also picks the window (radius 3 / radius 8)
        h->guard = h->d_small + 8;
        if (int rc = launch_sweep_y(h, (const int16_t*)h->zfield.ptr, zy_out, zy_side, nx, ny, nz, s, window_choice ? 3 : 0)) return rc;
        if (window_choice) {
            h->guard = h->d_small + 9;
            if (int rc = launch_sweep_y(h, (const int16_t*)h->zfield.ptr, zy_out, zy_side, nx, ny, nz, s, 8)) return rc;
        }
        h->guard = general_guard;
        launched_since_mark = true;
    } else if (fused) {
        if (int rc = launch_sweep_zy_fused(h, d_filled, zy_out, zy_side, nx, ny, nz, s)) return rc;
        launched_since_mark = true;
    } else {                                                    // (no far-field kernel for this shape / switched off: unbounded scans)
        if (int rc = launch_sweep_y(h, (const int16_t*)h->zfield.ptr, zy_out, zy_side, nx, ny, nz, s)) return rc;
        launched_since_mark = true;
    }
    HIP_TRY(h, mark(4));
    if (select) {
        if (int rc = launch_envelope(h, 2, (const int16_t*)h->zfield.ptr, nullptr, p16 ? h->plane16.ptr : nullptr, p16 ? (int32_t*)h->yzfield.ptr : nullptr,
                                     nx, ny, nz, resolution, vb, h->d_small, h->d_small + 4, s, 0, nullptr, ex2)) return rc;
        launched_since_mark = true;
    }
    HIP_TRY(h, mark(5));
    // x sweep: same choice
    if (!envelope || fused) h->far_y = nullptr;
    else h->far_y = h->d_small + 4;             // (K3/16 raises far_y + 1 = far_x)
    if (select) {
        DcDecide dx; dx.stage = 1; dx.dense_tried = dense; dx.handoff = handoff; dx.window_choice = window_choice;
        if (h->force_env < 0) {
            if (int rc = launch_envelope(h, 3, p16 ? (const int16_t*)h->plane16.ptr : nullptr, p16 ? (const int32_t*)h->yzfield.ptr : nullptr, d_out, nullptr,
                                         nx, ny, nz, resolution, vb, h->d_small, general_guard, s, 0, h->d_small + 12, ex3, &dx)) return rc;
        } else if (int rc = decide(1)) return rc;
        h->guard = h->d_small + 10;
        if (p16) {
            if (int rc = launch_sweep_x16(h, (const int16_t*)h->plane16.ptr, (const int32_t*)h->yzfield.ptr, d_out, 0, nx, 0,
                                          0, nx, ny, nz, 0, 0, 0, nx, resolution, vb, h->d_small, h->d_small + 2, s)) return rc;
        } else {
            if (int rc = launch_sweep_x(h, (const int32_t*)h->yzfield.ptr, d_out, 0, nx, 0, ny, nz, 0, 0, 0, nx,
                                        resolution, vb, h->d_small, h->d_small + 2, s, 0, -1, h->scan_x, h->far_y + 1)) return rc;
        }
        h->guard = general_guard;
        launched_since_mark = true;
    } else if (p16) {
        if (int rc = launch_sweep_x16(h, (const int16_t*)h->plane16.ptr, (const int32_t*)h->yzfield.ptr, d_out, 0, nx, 0,
                                      0, nx, ny, nz, 0, 0, 0, nx, resolution, vb, h->d_small, h->d_small + 2, s)) return rc;
        launched_since_mark = true;
    } else {
        if (int rc = launch_sweep_x(h, (const int32_t*)h->yzfield.ptr, d_out, 0, nx, 0, ny, nz, 0, 0, 0, nx,
                                    resolution, vb, h->d_small, h->d_small + 2, s)) return rc;
        launched_since_mark = true;
    }
    HIP_TRY(h, mark(6));
    if (select) {
        if (int rc = launch_envelope(h, 3, p16 ? (const int16_t*)h->plane16.ptr : nullptr, p16 ? (const int32_t*)h->yzfield.ptr : nullptr, d_out, nullptr,
                                     nx, ny, nz, resolution, vb, h->d_small, h->d_small + 5, s, 0, nullptr, ex3)) return rc;
        launched_since_mark = true;
    }
    }   // (!standby)
    // one kernel folds the maxima, publishes the status block (device copy for get_extrema, pinned host copy for the
    // next build's policy) and clears it for the next build
    // The report (status block -> pinned host memory + an event) is only taken when the previous report has been
    // consumed: the event is then never re-recorded while the host still waits for it, so even a caller that
    // enqueues builds back to back without ever synchronising keeps feeding the policy (a few builds late), and the
    // builds in between skip the host write.  What the reported build was (dense? fix-up? envelope mode?) is
    // remembered with the report -- not with whatever build happens to be the latest when it is read.
    // (only a build that carried the dense tier has anything to teach the policy: the builds of a pause leave the report slot
    //  free, so that the PROBE at its end is the build whose verdict comes back -- with every build reporting, the slot was
    //  usually taken by a paused build when the probe came, its failure went unseen and the pause never grew; tests/policy_harness)
    // (a stand-by build: the x sweep's launch has done it -- one launch less behind the dense kernel)
    // (a build without the dense tier reports to the second slot: its far flags teach far_predict)
    // (... the builds that probed: a predicted build's flags say nothing new, and a report is PCIe writes the fold waits for)
    const bool report_far = !report && select && !predicted && h->h_far_dev && !h->far_pending;
    if (!folded) if (int rc = fold_slots(h, h->d_small, s, h->d_result, report ? h->h_flags_dev : report_far ? h->h_far_dev : nullptr,
                                         report ? kReportDense : kReportFar)) return rc;
    if (report_far) {
        HIP_TRY(h, hipEventRecord(h->far_ev, s));
        h->far_pending = true;
    }
    h->small_clean = true;
    h->guard = nullptr;
    h->far_y = nullptr;
    if (report) {
        HIP_TRY(h, hipEventRecord(h->flags_ev, s));
        h->flags_pending = true;
        h->pol.prev = ReportedBuild{dense, dense_generic, cur_fix_mode, cur_staged, predicted};
    }
    if (prof) {
        HIP_TRY(h, mark(7));
        for (auto e : ev) h->events.push_back(e);
    }
    HIP_TRY(h, hipEventRecord(h->build_done_ev, s));
    h->order_valid = true; h->order_stream = s;
    h->last_stream = s;
    h->last_resolution = resolution;
    h->last_n = n;
    h->have_result = true;
    return SDFGPU_OK;
}

// Device -> pageable host memory at PCIe rate, for destinations nobody has touched yet (the std::vector / numpy array
// the reference API returns is fresh: 512^3 floats are 131 072 first-touch page faults, ~40 ms on one thread, and a plain
// hipMemcpy takes them one by one on the runtime's copy thread).  The DMA lands in two pinned staging chunks of the
// context; a small team of host threads copies each chunk out while the next one is in flight, so the faults are spread
// over the team and overlap the transfer.  Enqueued on `st` (ordered after the work already there).  (Round 1 pre-touched the destination from threads running beside the upload:
// the faults and the runtime's pinning of the source pages fought over the address space, and the 128 MiB upload took
// 29 ms instead of 2.5.)  Returns when the data is in dst.
constexpr size_t kPinChunk = (size_t)32 << 20;
constexpr size_t kPinMin = (size_t)4 << 20;      // below this a plain (runtime-staged) copy is as fast as the team

int copy_to_host(sdfgpu_handle h, void* dst, const void* d_src, size_t bytes, hipStream_t st = nullptr) {
    if (bytes == 0) return SDFGPU_OK;
    for (int i = 0; i < 2 && bytes >= kPinMin; ++i) {
        if (!h->pin[i] && hipHostMalloc(&h->pin[i], kPinChunk, hipHostMallocDefault) != hipSuccess) h->pin[i] = nullptr;
        if (h->pin[i] && !h->pin_ev[i] && hipEventCreateWithFlags(&h->pin_ev[i], hipEventDisableTiming) != hipSuccess) h->pin_ev[i] = nullptr;
    }
    if (bytes < kPinMin || !h->pin[0] || !h->pin[1] || !h->pin_ev[0] || !h->pin_ev[1]) {
        HIP_TRY(h, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipStreamSynchronize(st));
        return SDFGPU_OK;
    }
    {   // huge pages on the 2 MiB-aligned interior, where the kernel grants them: one fault then maps 512 small pages' worth
        constexpr uintptr_t kHuge = (uintptr_t)2 << 20;
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(dst) + kHuge - 1) & ~(kHuge - 1);
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(dst) + bytes) & ~(kHuge - 1);
        if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);
    }
    if (!h->team) h->team = new (std::nothrow) HostTeam();
    if (!h->team) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "out of host memory");
    // (the scheduler is host-only code, tests/sched_harness.cpp runs it under -fsanitize=thread: sdfgpu_hostteam.hpp)
    hipError_t herr = hipSuccess;
    const int rc = staged_drain(
        *h->team, bytes, kPinChunk, host_team_size(16),
        [&](int64_t i, int buf, size_t len) -> int {
            herr = hipMemcpyAsync(h->pin[buf], static_cast<const char*>(d_src) + (size_t)i * kPinChunk, len, hipMemcpyDeviceToHost, st);
            if (herr == hipSuccess) herr = hipEventRecord(h->pin_ev[buf], st);
            return herr == hipSuccess ? 0 : 1;
        },
        [&](int buf) -> int { herr = hipEventSynchronize(h->pin_ev[buf]); return herr == hipSuccess ? 0 : 1; },
        [&](int buf, size_t off, size_t goff, size_t len) { memcpy(static_cast<char*>(dst) + goff, static_cast<const char*>(h->pin[buf]) + off, len); });
    if (rc != 0) {
        (void)hipDeviceSynchronize();
        return fail(h, SDFGPU_ERR_HIP, "HIP error %d (%s) in the device-to-host copy", (int)herr, hipGetErrorString(herr));
    }
    return SDFGPU_OK;
}

// Pageable host -> device, the mirror image of copy_to_host: a synchronous hipMemcpy from pageable memory goes through
// the runtime's staging copy on one host thread (measured: 128 MiB in 24 - 30 ms, i.e. 5 GB/s, where the link does 52);
// here a team of threads fills the two pinned chunks and the DMA of one chunk overlaps the filling of the next.
// Enqueued on `st`; returns when the last chunk's DMA has completed.
// `fill(dst, out_offset, len)` produces bytes [out_offset, out_offset + len) of the upload into dst (slices are multiples
// of 4096 bytes except the last): a plain memcpy for copy_from_host, the cells / mask -> bits classification for the
// host-buffer builds (round 5), which is what makes the team worth more than a copy -- it reads 8 or 64 bytes per byte sent.
template <class Fill>
int staged_upload_hip(sdfgpu_handle h, void* d_dst, size_t bytes, hipStream_t st, int team_cap, Fill fill) {
    if (bytes == 0) return SDFGPU_OK;
    for (int i = 0; i < 2; ++i) {
        if (!h->pin[i] && hipHostMalloc(&h->pin[i], kPinChunk, hipHostMallocDefault) != hipSuccess) h->pin[i] = nullptr;
        if (h->pin[i] && !h->pin_ev[i] && hipEventCreateWithFlags(&h->pin_ev[i], hipEventDisableTiming) != hipSuccess) h->pin_ev[i] = nullptr;
    }
    if (!h->pin[0] || !h->pin[1] || !h->pin_ev[0] || !h->pin_ev[1]) {
        // no pinned staging (a locked-memory limit): produce the bytes into a pageable temporary, one runtime-staged copy each
        // (slower, never wrong -- ADVICE r5: rounds 1 - 4 had this fallback, round 5 had dropped it)
        (void)hipGetLastError();
        std::vector<char> tmp(std::min(bytes, kPinChunk));
        for (size_t o = 0; o < bytes; o += tmp.size()) {
            const size_t len = std::min(tmp.size(), bytes - o);
            fill(tmp.data(), o, len);
            HIP_TRY(h, hipMemcpyAsync(static_cast<char*>(d_dst) + o, tmp.data(), len, hipMemcpyHostToDevice, st));
            HIP_TRY(h, hipStreamSynchronize(st));           // (tmp is refilled by the next round)
        }
        return SDFGPU_OK;
    }
    if (!h->team) h->team = new (std::nothrow) HostTeam();
    if (!h->team) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "out of host memory");
    hipError_t herr = hipSuccess;
    const int rc = sdfgpu::staged_upload(
        *h->team, bytes, kPinChunk, host_team_size(team_cap),
        [&](int buf, size_t off, size_t goff, size_t len) { fill(static_cast<char*>(h->pin[buf]) + off, goff, len); },
        [&](int64_t i, int buf, size_t len) -> int {
            herr = hipMemcpyAsync(static_cast<char*>(d_dst) + (size_t)i * kPinChunk, h->pin[buf], len, hipMemcpyHostToDevice, st);
            if (herr == hipSuccess) herr = hipEventRecord(h->pin_ev[buf], st);
            return herr == hipSuccess ? 0 : 1;
        },
        [&](int buf) -> int { herr = hipEventSynchronize(h->pin_ev[buf]); return herr == hipSuccess ? 0 : 1; });
    if (rc != 0) {
        (void)hipDeviceSynchronize();
        return fail(h, SDFGPU_ERR_HIP, "HIP error %d (%s) in the host-to-device copy", (int)herr, hipGetErrorString(herr));
    }
    return SDFGPU_OK;
}

int copy_from_host(sdfgpu_handle h, void* d_dst, const void* src, size_t bytes, hipStream_t st = nullptr) {
    if (bytes == 0) return SDFGPU_OK;
    if (bytes < kPinMin) {
        HIP_TRY(h, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipStreamSynchronize(st));
        return SDFGPU_OK;
    }
    return staged_upload_hip(h, d_dst, bytes, st, 16, [src](char* dst, size_t o, size_t len) { memcpy(dst, static_cast<const char*>(src) + o, len); });
}

// ---- host-side classification: cells / mask -> one bit per voxel (round 5) ----------------------------------
// Bit (v & 7) of byte (v >> 3) = voxel v is filled, with the predicates of the device classifiers: mask byte != 0;
// occupancy > 0.5f || (unknown_is_filled && occupancy == 0.5f) (collision_map.hpp:689-704 -- the reference compares the float
// with the double 0.5, which is the float 0.5f exactly; NaN is "free" in both).  out[0 .. len) covers voxels
// [8 ob, 8 (ob + len)); bits of voxels >= n are 0.
void pack_mask_bits(const uint8_t* m, int64_t n, size_t ob, size_t len, uint8_t* out) {
    const int64_t v0 = (int64_t)ob * 8, v1 = std::min<int64_t>(n, v0 + (int64_t)len * 8);
    int64_t v = v0;
    size_t o = 0;
#if defined(__SSE2__)
    const __m128i zero = _mm_setzero_si128();
    for (; v + 16 <= v1; v += 16, o += 2) {
        const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i*>(m + v));
        const uint32_t nz = ~(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(x, zero)) & 0xffffu;
        out[o] = (uint8_t)nz; out[o + 1] = (uint8_t)(nz >> 8);
    }
#else       // (hosts without SSE2: eight bytes -> eight bits by the multiply trick)
    for (; v + 8 <= v1; v += 8, ++o) {
        uint64_t x;
        memcpy(&x, m + v, 8);
        x = (x | ((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full)) & 0x8080808080808080ull;
        out[o] = (uint8_t)(((x >> 7) * 0x0102040810204080ull) >> 56);
    }
#endif
    for (; o < len; ++o) {
        uint32_t b = 0;
        for (int k = 0; k < 8; ++k, ++v) if (v < v1 && m[v]) b |= 1u << k;
        out[o] = (uint8_t)b;
    }
}
void pack_cells_bits(const char* cells, size_t stride, size_t off, int unknown, int64_t n, size_t ob, size_t len, uint8_t* out) {
    const int64_t v0 = (int64_t)ob * 8, v1 = std::min<int64_t>(n, v0 + (int64_t)len * 8);
    int64_t v = v0;
    size_t o = 0;
#if defined(__SSE2__)
    if (stride == 8 && (off == 0 || off == 4)) {                // COLLISION_CELL {float occupancy; uint32 component}: 4 records per pair of loads
        const __m128 half = _mm_set1_ps(0.5f);
        const __m128 unk = _mm_castsi128_ps(_mm_set1_epi32(unknown ? -1 : 0));
        const char* p = cells + (size_t)v0 * 8;
        for (; v + 8 <= v1; v += 8, ++o, p += 64) {
            uint32_t b = 0;
            for (int g = 0; g < 2; ++g) {
                const __m128 a0 = _mm_loadu_ps(reinterpret_cast<const float*>(p + 32 * g));
                const __m128 a1 = _mm_loadu_ps(reinterpret_cast<const float*>(p + 32 * g + 16));
                const __m128 occ = off == 0 ? _mm_shuffle_ps(a0, a1, _MM_SHUFFLE(2, 0, 2, 0)) : _mm_shuffle_ps(a0, a1, _MM_SHUFFLE(3, 1, 3, 1));
                const __m128 f = _mm_or_ps(_mm_cmpgt_ps(occ, half), _mm_and_ps(_mm_cmpeq_ps(occ, half), unk));
                b |= (uint32_t)_mm_movemask_ps(f) << (4 * g);
            }
            out[o] = (uint8_t)b;
        }
    }
#endif
    for (; o < len; ++o) {
        uint32_t b = 0;
        for (int k = 0; k < 8; ++k, ++v) {
            if (v >= v1) continue;
            float occ;
            memcpy(&occ, cells + (size_t)v * stride + off, 4);
            if (occ > 0.5f || (unknown && occ == 0.5f)) b |= 1u << k;
        }
        out[o] = (uint8_t)b;
    }
}

// Host cells / mask -> bits (team) -> device -> byte mask at d_mask_dst (n bytes), enqueued on `st`; the host buffer has been
// consumed when this returns (the unpack kernel is stream-ordered behind the upload).
int upload_packed(sdfgpu_handle h, const uint8_t* filled, const void* cells, size_t stride, size_t off, int unknown, int64_t n,
                  uint8_t* d_mask_dst, hipStream_t st) {
    const size_t nbytes = ((size_t)n + 7) / 8, padded = (nbytes + 3) & ~(size_t)3;
    if (int rc = ensure(h, h->stage_bits, padded, "bit staging")) return rc;
    auto fill = [=](char* dst, size_t o, size_t len) {
        const size_t real = o >= nbytes ? 0 : std::min(len, nbytes - o);
        if (real) {
            if (cells) pack_cells_bits(static_cast<const char*>(cells), stride, off, unknown, n, o, real, reinterpret_cast<uint8_t*>(dst));
            else pack_mask_bits(filled, n, o, real, reinterpret_cast<uint8_t*>(dst));
        }
        if (real < len) memset(dst + real, 0, len - real);
    };
    if (padded >= ((size_t)256 << 10)) {
        if (int rc = staged_upload_hip(h, h->stage_bits.ptr, padded, st, 32, fill)) return rc;
    } else {                                                    // small grids: one thread, one runtime-staged copy
        std::vector<char> tmp(padded);
        fill(tmp.data(), 0, padded);
        HIP_TRY(h, hipMemcpyAsync(h->stage_bits.ptr, tmp.data(), padded, hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipStreamSynchronize(st));                   // (tmp goes out of scope)
    }
    if (!d_mask_dst) return SDFGPU_OK;                          // (the host-buffer builds: the bits are the build's input)
    const int64_t nch = (n + 15) / 16;
    hipLaunchKernelGGL(k_unpack_bits_mask, dim3((unsigned)((nch + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       (const uint32_t*)h->stage_bits.ptr, d_mask_dst, n);
    HIP_TRY(h, hipGetLastError());
    return SDFGPU_OK;
}

// the caller's host bit field (ceil(n / 32) words) -> h->stage_bits
int upload_bits(sdfgpu_handle h, const uint32_t* bits, int64_t n, hipStream_t st) {
    const size_t bytes = (((size_t)n + 31) / 32) * 4;
    if (int rc = ensure(h, h->stage_bits, bytes, "bit staging")) return rc;
    return copy_from_host(h, h->stage_bits.ptr, bits, bytes, st);
}

// d_out_user != nullptr: the field stays on the device in the caller's buffer (no download; out_sdf is unused)
// bits_in != nullptr: the caller's occupancy is already one bit per voxel (sdfgpu_build_bits)
int build_host_impl(sdfgpu_handle h, const uint8_t* filled, const void* cells, size_t stride, size_t off,
                    int unknown, int64_t nx, int64_t ny, int64_t nz, double resolution, int vb, float* out_sdf,
                    double* out_max, double* out_min, float* d_out_user = nullptr, const uint32_t* bits_in = nullptr) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if ((!filled && !cells && !bits_in) || (!out_sdf && !d_out_user)) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null pointer");
    if (int rc = check_dims(h, nx, ny, nz)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const int64_t n = nx * ny * nz;
    const size_t in_bytes = cells ? (size_t)n * stride : (size_t)n;
    h->tag_cached_bytes = 0;                        // (stage_in is about to be overwritten)
    // Round 5: the host's thread team classifies the caller's buffer into one bit per voxel while it fills the pinned staging
    // chunk, and 1/8 B per voxel crosses PCIe (16 MiB at 512^3 instead of 128 MiB of mask or 1 GiB of COLLISION_CELL records);
    // k_unpack_bits_mask spreads them into the byte mask on the device.  Same predicate as the device classifier
    // (pack_cells_bits); "host_pack" = 0 keeps the upload-and-classify-on-device path (device-resident cells always take it).
    const bool packed = !bits_in && (h->host_pack == 2 || (h->host_pack == 1 && in_bytes >= kPinMin));
    if (!packed && !bits_in) if (int rc = ensure(h, h->stage_in, in_bytes, "input staging")) return rc;
    if (!d_out_user) if (int rc = ensure(h, h->stage_out, (size_t)n * 4, "output staging")) return rc;
    float* const d_out = d_out_user ? d_out_user : (float*)h->stage_out.ptr;
    const bool timing = getenv("SDFGPU_HOST_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t1 = now();
    // (out_sdf is scratch from here on: include/sdfgpu.h documents that its contents are undefined when the call fails)
    // (round 6: the uploaded bits ARE the build's input -- no byte mask is spread out on the device and packed again by K0)
    if (packed) { if (int rc0 = upload_packed(h, filled, cells, stride, off, unknown, n, nullptr, nullptr)) return rc0; }
    else if (bits_in) { if (int rc0 = upload_bits(h, bits_in, n, nullptr)) return rc0; }
    else if (int rc0 = copy_from_host(h, h->stage_in.ptr, cells ? cells : (const void*)filled, in_bytes)) return rc0;
    const double t2 = now();
    int rc = (packed || bits_in) ? build_device_impl(h, nullptr, nullptr, 0, 0, 0, nx, ny, nz, resolution, vb, d_out, nullptr, (const uint32_t*)h->stage_bits.ptr)
                    : build_device_impl(h, cells ? nullptr : (const uint8_t*)h->stage_in.ptr,
                                        cells ? h->stage_in.ptr : nullptr, stride, off, unknown, nx, ny, nz, resolution,
                                        vb, d_out, nullptr);
    if (rc) return rc;
    const double t3 = now();
    if (!d_out_user) if (int rc2 = copy_to_host(h, out_sdf, h->stage_out.ptr, (size_t)n * 4)) return rc2;
    const double t5 = now();
    double mx, mn;
    rc = sdfgpu_get_extrema(h, &mx, &mn);
    if (timing) fprintf(stderr, "[sdfgpu host] h2d %.3f enqueue %.3f build + d2h %.3f extrema %.3f ms\n", t2 - t1, t3 - t2, t5 - t3, now() - t5);
    if (rc) return rc;
    if (out_max) *out_max = mx;
    if (out_min) *out_min = mn;
    return SDFGPU_OK;
}

}  // namespace

extern "C" {

const char* sdfgpu_version(void) { return "sdfgpu 0.1 (gfx950, HIP)"; }

int sdfgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sdfgpu_create(int device, sdfgpu_handle* out_handle) {
    if (!out_handle) return fail(nullptr, SDFGPU_ERR_INVALID_ARGUMENT, "out_handle is null");
    *out_handle = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, SDFGPU_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n)
        return fail(nullptr, SDFGPU_ERR_NO_DEVICE, "device index %d out of range (0..%d)", device, n - 1);
    HIP_TRY(nullptr, hipSetDevice(device));
    sdfgpu_context* ctx = new (std::nothrow) sdfgpu_context();
    if (!ctx) return fail(nullptr, SDFGPU_ERR_INVALID_ARGUMENT, "out of host memory");
    ctx->device = device;
    {
        const char* rz = getenv("SDFGPU_REDZONE");
        ctx->redzone = rz && rz[0] && rz[0] != '0';
    }
    hipError_t ce = rz_malloc(ctx, "extrema slots", (size_t)kSlots * kSlotWords * 4, (void**)&ctx->d_slots) == SDFGPU_OK ? hipSuccess : hipErrorOutOfMemory;
    if (ce == hipSuccess) ce = hipMemset(ctx->d_slots, 0, (size_t)kSlots * kSlotWords * 4);
    if (ce == hipSuccess) ce = rz_malloc(ctx, "status block", 512, (void**)&ctx->d_small) == SDFGPU_OK ? hipSuccess : hipErrorOutOfMemory;
    if (ce == hipSuccess) ce = hipMemset(ctx->d_small, 0, 512);
    if (ce == hipSuccess) ce = hipEventCreateWithFlags(&ctx->build_done_ev, hipEventDisableTiming);
    if (ce != hipSuccess) {
        if (ctx->d_slots) (void)rz_free(ctx, ctx->d_slots);
        if (ctx->d_small) (void)rz_free(ctx, ctx->d_small);
        if (ctx->build_done_ev) (void)hipEventDestroy(ctx->build_done_ev);
        delete ctx;
        return fail(nullptr, SDFGPU_ERR_HIP, "HIP error %d (%s) while allocating the context's status blocks",
                    (int)ce, hipGetErrorString(ce));
    }
    ctx->d_result = ctx->d_small + 64;                       // second half of the same allocation
    if (hipHostMalloc((void**)&ctx->h_flags, 256, hipHostMallocMapped) != hipSuccess) ctx->h_flags = nullptr;
    if (ctx->h_flags && hipHostGetDevicePointer((void**)&ctx->h_flags_dev, ctx->h_flags, 0) != hipSuccess) ctx->h_flags_dev = nullptr;
    if (ctx->h_flags && hipEventCreateWithFlags(&ctx->flags_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipHostFree(ctx->h_flags);
        ctx->h_flags = nullptr;
    }
    if (ctx->h_flags && ctx->h_flags_dev && hipEventCreateWithFlags(&ctx->far_ev, hipEventDisableTiming) == hipSuccess) {
        memset(ctx->h_flags, 0, 256);
        ctx->h_far = ctx->h_flags + 32;                        // second half of the same pinned allocation
        ctx->h_far_dev = ctx->h_flags_dev + 32;
    } else {
        ctx->far_ev = nullptr;
    }
    *out_handle = ctx;
    return SDFGPU_OK;
}

int sdfgpu_destroy(sdfgpu_handle h) {
    if (!h) return SDFGPU_OK;
    (void)hipSetDevice(h->device);
    for (DeviceBuffer* b : {&h->zfield, &h->yzfield, &h->plane16, &h->bits, &h->unc, &h->tileflag, &h->fix_order, &h->tagmask, &h->tagids, &h->stage_in,
                            &h->stage_bits, &h->stage_out, &h->query_stage, &h->planebits})
        if (b->ptr) (void)rz_free(h, b->ptr);
    if (h->d_small) (void)rz_free(h, h->d_small);
    if (h->d_slots) (void)rz_free(h, h->d_slots);
    if (h->rz_table) (void)hipFree(h->rz_table);
    if (h->h_flags) { (void)hipHostFree(h->h_flags); (void)hipEventDestroy(h->flags_ev); }
    if (h->far_ev) (void)hipEventDestroy(h->far_ev);
    if (h->build_done_ev) (void)hipEventDestroy(h->build_done_ev);
    for (int i = 0; i < 2; ++i) { if (h->pin[i]) (void)hipHostFree(h->pin[i]); if (h->pin_ev[i]) (void)hipEventDestroy(h->pin_ev[i]); }
    for (size_t i = 0; i < h->events.size(); ++i)
        if (i % 8 == 0 || h->events[i] != h->events[i - 1]) (void)hipEventDestroy(h->events[i]);
    for (hipEvent_t e : h->event_pool) (void)hipEventDestroy(e);
    delete h->team;                                             // (joins the parked host threads)
    delete h;
    return SDFGPU_OK;
}

const char* sdfgpu_last_error(sdfgpu_handle h) { return h ? h->error.c_str() : g_create_error.c_str(); }

static int sdfgpu_build_body(sdfgpu_handle h, const uint8_t* filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                 int add_virtual_border, float* out_sdf, double* out_max, double* out_min) {
    return build_host_impl(h, filled, nullptr, 0, 0, 0, nx, ny, nz, resolution, add_virtual_border, out_sdf,
                           out_max, out_min);
}
int sdfgpu_build(sdfgpu_handle h, const uint8_t* filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                 int add_virtual_border, float* out_sdf, double* out_max, double* out_min) {
    return rz_wrap(h, nullptr, [&]() -> int { return sdfgpu_build_body(h, filled, nx, ny, nz, resolution, add_virtual_border, out_sdf, out_max, out_min); });
}

static int sdfgpu_build_cells_body(sdfgpu_handle h, const void* cells, size_t cell_stride, size_t occupancy_offset,
                       int unknown_is_filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                       int add_virtual_border, float* out_sdf, double* out_max, double* out_min) {
    if (h && !cells) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "cells is null");
    if (h && (cell_stride < 4 || (cell_stride % 4) || (occupancy_offset % 4) || occupancy_offset + 4 > cell_stride))
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "cell_stride/occupancy_offset must be 4-byte aligned and in range");
    return build_host_impl(h, nullptr, cells, cell_stride, occupancy_offset, unknown_is_filled, nx, ny, nz,
                           resolution, add_virtual_border, out_sdf, out_max, out_min);
}
int sdfgpu_build_cells(sdfgpu_handle h, const void* cells, size_t cell_stride, size_t occupancy_offset,
                       int unknown_is_filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                       int add_virtual_border, float* out_sdf, double* out_max, double* out_min) {
    return rz_wrap(h, nullptr, [&]() -> int { return sdfgpu_build_cells_body(h, cells, cell_stride, occupancy_offset, unknown_is_filled, nx, ny, nz, resolution, add_virtual_border, out_sdf, out_max, out_min); });
}

static int sdfgpu_build_to_device_body(sdfgpu_handle h, const uint8_t* filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                           int add_virtual_border, float* d_out_sdf, double* out_max, double* out_min) {
    if (h && !d_out_sdf) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "d_out_sdf is null");
    return build_host_impl(h, filled, nullptr, 0, 0, 0, nx, ny, nz, resolution, add_virtual_border, nullptr, out_max, out_min,
                           d_out_sdf);
}
int sdfgpu_build_to_device(sdfgpu_handle h, const uint8_t* filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                           int add_virtual_border, float* d_out_sdf, double* out_max, double* out_min) {
    return rz_wrap(h, nullptr, [&]() -> int { return sdfgpu_build_to_device_body(h, filled, nx, ny, nz, resolution, add_virtual_border, d_out_sdf, out_max, out_min); });
}

static int sdfgpu_build_cells_to_device_body(sdfgpu_handle h, const void* cells, size_t cell_stride, size_t occupancy_offset,
                                 int unknown_is_filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                                 int add_virtual_border, float* d_out_sdf, double* out_max, double* out_min) {
    if (h && !cells) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "cells is null");
    if (h && !d_out_sdf) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "d_out_sdf is null");
    if (h && (cell_stride < 4 || (cell_stride % 4) || (occupancy_offset % 4) || occupancy_offset + 4 > cell_stride))
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "cell_stride/occupancy_offset must be 4-byte aligned and in range");
    return build_host_impl(h, nullptr, cells, cell_stride, occupancy_offset, unknown_is_filled, nx, ny, nz, resolution,
                           add_virtual_border, nullptr, out_max, out_min, d_out_sdf);
}
int sdfgpu_build_cells_to_device(sdfgpu_handle h, const void* cells, size_t cell_stride, size_t occupancy_offset,
                                 int unknown_is_filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                                 int add_virtual_border, float* d_out_sdf, double* out_max, double* out_min) {
    return rz_wrap(h, nullptr, [&]() -> int { return sdfgpu_build_cells_to_device_body(h, cells, cell_stride, occupancy_offset, unknown_is_filled, nx, ny, nz, resolution, add_virtual_border, d_out_sdf, out_max, out_min); });
}

static int sdfgpu_build_device_body(sdfgpu_handle h, const uint8_t* d_filled, int64_t nx, int64_t ny, int64_t nz,
                        double resolution, int add_virtual_border, float* d_out_sdf, void* stream) {
    if (h && !d_filled) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "d_filled is null");
    return build_device_impl(h, d_filled, nullptr, 0, 0, 0, nx, ny, nz, resolution, add_virtual_border, d_out_sdf,
                             (hipStream_t)stream);
}
int sdfgpu_build_device(sdfgpu_handle h, const uint8_t* d_filled, int64_t nx, int64_t ny, int64_t nz,
                        double resolution, int add_virtual_border, float* d_out_sdf, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_build_device_body(h, d_filled, nx, ny, nz, resolution, add_virtual_border, d_out_sdf, stream); });
}

static int sdfgpu_build_cells_device_body(sdfgpu_handle h, const void* d_cells, size_t cell_stride, size_t occupancy_offset,
                              int unknown_is_filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                              int add_virtual_border, float* d_out_sdf, void* stream) {
    if (h && !d_cells) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "d_cells is null");
    return build_device_impl(h, nullptr, d_cells, cell_stride, occupancy_offset, unknown_is_filled, nx, ny, nz,
                             resolution, add_virtual_border, d_out_sdf, (hipStream_t)stream);
}
int sdfgpu_build_cells_device(sdfgpu_handle h, const void* d_cells, size_t cell_stride, size_t occupancy_offset,
                              int unknown_is_filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                              int add_virtual_border, float* d_out_sdf, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_build_cells_device_body(h, d_cells, cell_stride, occupancy_offset, unknown_is_filled, nx, ny, nz, resolution, add_virtual_border, d_out_sdf, stream); });
}

static int sdfgpu_build_bits_device_body(sdfgpu_handle h, const uint32_t* d_bits, int64_t nx, int64_t ny, int64_t nz,
                             double resolution, int add_virtual_border, float* d_out_sdf, void* stream) {
    if (h && !d_bits) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "d_bits is null");
    return build_device_impl(h, nullptr, nullptr, 0, 0, 0, nx, ny, nz, resolution, add_virtual_border, d_out_sdf,
                             (hipStream_t)stream, d_bits);
}
int sdfgpu_build_bits_device(sdfgpu_handle h, const uint32_t* d_bits, int64_t nx, int64_t ny, int64_t nz,
                             double resolution, int add_virtual_border, float* d_out_sdf, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_build_bits_device_body(h, d_bits, nx, ny, nz, resolution, add_virtual_border, d_out_sdf, stream); });
}

static int sdfgpu_build_bits_body(sdfgpu_handle h, const uint32_t* bits, int64_t nx, int64_t ny, int64_t nz, double resolution,
                      int add_virtual_border, float* out_sdf, double* out_max, double* out_min) {
    if (h && !bits) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "bits is null");
    return build_host_impl(h, nullptr, nullptr, 0, 0, 0, nx, ny, nz, resolution, add_virtual_border, out_sdf, out_max, out_min,
                           nullptr, bits);
}
int sdfgpu_build_bits(sdfgpu_handle h, const uint32_t* bits, int64_t nx, int64_t ny, int64_t nz, double resolution,
                      int add_virtual_border, float* out_sdf, double* out_max, double* out_min) {
    return rz_wrap(h, nullptr, [&]() -> int { return sdfgpu_build_bits_body(h, bits, nx, ny, nz, resolution, add_virtual_border, out_sdf, out_max, out_min); });
}

static int sdfgpu_copy_to_host_body(sdfgpu_handle h, void* dst, const void* d_src, size_t bytes, void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (bytes > 0 && (!dst || !d_src)) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    return copy_to_host(h, dst, d_src, bytes, (hipStream_t)stream);
}
int sdfgpu_copy_to_host(sdfgpu_handle h, void* dst, const void* d_src, size_t bytes, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_copy_to_host_body(h, dst, d_src, bytes, stream); });
}

static int sdfgpu_copy_from_host_body(sdfgpu_handle h, void* d_dst, const void* src, size_t bytes, void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (bytes > 0 && (!d_dst || !src)) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    return copy_from_host(h, d_dst, src, bytes, (hipStream_t)stream);
}
int sdfgpu_copy_from_host(sdfgpu_handle h, void* d_dst, const void* src, size_t bytes, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_copy_from_host_body(h, d_dst, src, bytes, stream); });
}

static int sdfgpu_upload_classified_body(sdfgpu_handle h, const uint8_t* filled, const void* cells, size_t cell_stride, size_t occupancy_offset,
                             int unknown_is_filled, int64_t n, uint8_t* d_mask, void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if ((!filled && !cells) || !d_mask) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null pointer");
    if (n <= 0) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "voxel count must be positive");
    if (cells && (cell_stride < 4 || (cell_stride % 4) || (occupancy_offset % 4) || occupancy_offset + 4 > cell_stride))
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "cell_stride/occupancy_offset must be 4-byte aligned and in range");
    HIP_TRY(h, hipSetDevice(h->device));
    return upload_packed(h, cells ? nullptr : filled, cells, cell_stride, occupancy_offset, unknown_is_filled, n, d_mask, (hipStream_t)stream);
}
int sdfgpu_upload_classified(sdfgpu_handle h, const uint8_t* filled, const void* cells, size_t cell_stride, size_t occupancy_offset,
                             int unknown_is_filled, int64_t n, uint8_t* d_mask, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_upload_classified_body(h, filled, cells, cell_stride, occupancy_offset, unknown_is_filled, n, d_mask, stream); });
}

int sdfgpu_extrema_from_dsq(uint32_t max_dsq_free, uint32_t max_dsq_filled, double resolution, double* out_max,
                            double* out_min) {
    const double inf = std::numeric_limits<double>::infinity();
    // sdf_generation.hpp:246-269: max is attained on a free voxel, min on a filled one;
    // an absent class leaves the running value at its -inf / +inf start or at the
    // other class's infinite distance (all-free -> (inf, inf); all-filled -> (-inf, -inf)).
    double mx, mn;
    if (max_dsq_free == 0) mx = -inf;
    else if (max_dsq_free >= (uint32_t)kInf32) mx = inf;
    else mx = std::sqrt((double)max_dsq_free) * resolution;
    if (max_dsq_filled == 0) mn = inf;
    else if (max_dsq_filled >= (uint32_t)kInf32) mn = -inf;
    else mn = 0.0 - std::sqrt((double)max_dsq_filled) * resolution;
    if (out_max) *out_max = mx;
    if (out_min) *out_min = mn;
    return SDFGPU_OK;
}

int sdfgpu_get_extrema(sdfgpu_handle h, double* out_max, double* out_min) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!h->have_result) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "no build has been issued on this handle");
    HIP_TRY(h, hipSetDevice(h->device));
    uint32_t v[4];
    // (the build's own event, not its stream: the stream is the caller's and may be gone by now -- ADVICE r4)
    if (h->order_valid) HIP_TRY(h, hipEventSynchronize(h->build_done_ev));
    HIP_TRY(h, hipMemcpy(v, h->d_result, sizeof v, hipMemcpyDeviceToHost));
    return sdfgpu_extrema_from_dsq(v[0], v[1], h->last_resolution, out_max, out_min);
}

static int sdfgpu_sweep_zy_tiered_device_body(sdfgpu_handle h, const uint8_t* d_filled, int64_t nxs, int64_t ny, int64_t nz,
                                  int32_t* d_plane_dsq, uint32_t* d_far, void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    h->guard = nullptr;
    h->far_y = nullptr;
    if (!d_filled || !d_plane_dsq) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null device pointer");
    if (int rc = check_dims(h, nxs, ny, nz)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const int64_t n = nxs * ny * nz;
    hipStream_t s = (hipStream_t)stream;
    // (this call uses the handle's status block and scratch fields like a whole build does: it takes part in the same
    //  cross-stream ordering -- waits for the handle's previous work when that ran on another stream, and is waited for)
    if (h->order_valid && s != h->order_stream) HIP_TRY(h, hipStreamWaitEvent(s, h->build_done_ev, 0));
    struct OrderEnd { sdfgpu_handle h; hipStream_t s; ~OrderEnd() { if (hipEventRecord(h->build_done_ev, s) == hipSuccess) { h->order_valid = true; h->order_stream = s; } } } order_end{h, s};
    const bool tiered = h->envelope_on && far_geometry_ok(h, 2, nxs, ny, nz) &&
                        (nz % 4) == 0 && (reinterpret_cast<uintptr_t>(d_plane_dsq) % 16) == 0;
    if (!tiered) {
        if (fused_zy_eligible(h, d_filled, d_plane_dsq, nz)) return launch_sweep_zy_fused(h, d_filled, d_plane_dsq, nullptr, nxs, ny, nz, s);
        if (int rc = ensure(h, h->zfield, (size_t)n * 2, "z field")) return rc;
        if (int rc = launch_sweep_z(h, d_filled, nullptr, 0, 0, 0, nxs, ny, nz, (int16_t*)h->zfield.ptr, s)) return rc;
        return launch_sweep_y(h, (const int16_t*)h->zfield.ptr, d_plane_dsq, nullptr, nxs, ny, nz, s);
    }
    // K1, then the y sweep picked on the device: probe -> decide -> marching (bounded scan) / envelope, int32 output
    if (int rc = ensure(h, h->zfield, (size_t)n * 2, "z field")) return rc;
    HIP_TRY(h, hipMemsetAsync(h->d_small, 0, 128, s));
    h->small_clean = false;
    if (int rc = launch_sweep_z(h, d_filled, nullptr, 0, 0, 0, nxs, ny, nz, (int16_t*)h->zfield.ptr, s)) return rc;
    DcExtra ex;
    ex.out_i32 = d_plane_dsq;
    DcDecide dy; dy.stage = 0;
    if (h->force_env < 0) {
        if (int rc = launch_envelope(h, 2, (const int16_t*)h->zfield.ptr, nullptr, nullptr, nullptr, nxs, ny, nz, 1.0, 0,
                                     h->d_small, nullptr, s, 0, h->d_small + 12, &ex, &dy)) return rc;
    } else if (int rc = launch_decide(h, 0, false, s)) return rc;
    h->guard = h->d_small + 8;
    h->far_y = h->d_small + 4;
    h->scan_y = kScanExpectNear;
    int rc = launch_sweep_y(h, (const int16_t*)h->zfield.ptr, d_plane_dsq, nullptr, nxs, ny, nz, s);
    h->guard = nullptr;
    h->far_y = nullptr;
    if (rc) return rc;
    if (int rc2 = launch_envelope(h, 2, (const int16_t*)h->zfield.ptr, nullptr, nullptr, nullptr, nxs, ny, nz, 1.0, 0,
                                  h->d_small, h->d_small + 4, s, 0, nullptr, &ex)) return rc2;
    if (d_far) HIP_TRY(h, hipMemcpyAsync(d_far, h->d_small + 4, 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(h, hipMemsetAsync(h->d_small, 0, 128, s));
    h->small_clean = true;
    return SDFGPU_OK;
}
int sdfgpu_sweep_zy_tiered_device(sdfgpu_handle h, const uint8_t* d_filled, int64_t nxs, int64_t ny, int64_t nz,
                                  int32_t* d_plane_dsq, uint32_t* d_far, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_sweep_zy_tiered_device_body(h, d_filled, nxs, ny, nz, d_plane_dsq, d_far, stream); });
}

static int sdfgpu_sweep_zy_device_body(sdfgpu_handle h, const uint8_t* d_filled, int64_t nxs, int64_t ny, int64_t nz,
                           int32_t* d_plane_dsq, void* stream) {
    return sdfgpu_sweep_zy_tiered_device(h, d_filled, nxs, ny, nz, d_plane_dsq, nullptr, stream);
}
int sdfgpu_sweep_zy_device(sdfgpu_handle h, const uint8_t* d_filled, int64_t nxs, int64_t ny, int64_t nz,
                           int32_t* d_plane_dsq, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_sweep_zy_device_body(h, d_filled, nxs, ny, nz, d_plane_dsq, stream); });
}

static int sdfgpu_sweep_x_lines_device_body(sdfgpu_handle h, const int32_t* d_plane_dsq, int64_t nx, int64_t nys, int64_t nz,
                                int64_t y_global, int64_t ny_global, double resolution, int add_virtual_border,
                                float* d_out_sdf, uint32_t* d_maxdsq, void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!d_plane_dsq || !d_out_sdf || !d_maxdsq) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null device pointer");
    if (y_global < 0 || y_global + nys > ny_global) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "inconsistent slab geometry");
    if (int rc = check_dims(h, nx, nys, nz)) return rc;
    if (int rc = check_dims(h, nx, ny_global, nz)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    h->guard = nullptr;
    h->far_y = nullptr;
    if (h->order_valid && s != h->order_stream) HIP_TRY(h, hipStreamWaitEvent(s, h->build_done_ev, 0));
    struct OrderEnd { sdfgpu_handle h; hipStream_t s; ~OrderEnd() { if (hipEventRecord(h->build_done_ev, s) == hipSuccess) { h->order_valid = true; h->order_stream = s; } } } order_end{h, s};
    const bool tiered = h->envelope_on && far_geometry_ok(h, 3, nx, nys, nz, ny_global) &&
                        (reinterpret_cast<uintptr_t>(d_plane_dsq) % 16) == 0;
    if (!tiered) {
        if (int rc = launch_sweep_x(h, d_plane_dsq, d_out_sdf, 0, nx, 0, nys, nz, 0, 0, 0, nx, resolution, add_virtual_border,
                                    d_maxdsq, nullptr, s, y_global, ny_global)) return rc;
        return h->defer_fold ? SDFGPU_OK : fold_slots(h, d_maxdsq, s);
    }
    HIP_TRY(h, hipMemsetAsync(h->d_small, 0, 128, s));
    h->small_clean = false;
    DcExtra ex;
    ex.in_i32 = d_plane_dsq; ex.y_off = y_global; ex.ny_glob = ny_global;
    DcDecide dx; dx.stage = 1;
    if (h->force_env < 0) {
        if (int rc = launch_envelope(h, 3, nullptr, nullptr, d_out_sdf, nullptr, nx, nys, nz, resolution, add_virtual_border,
                                     h->d_small, nullptr, s, 0, h->d_small + 12, &ex, &dx)) return rc;
    } else if (int rc = launch_decide(h, 1, false, s)) return rc;
    h->guard = h->d_small + 10;
    int rc = launch_sweep_x(h, d_plane_dsq, d_out_sdf, 0, nx, 0, nys, nz, 0, 0, 0, nx, resolution, add_virtual_border,
                            d_maxdsq, nullptr, s, y_global, ny_global, kScanExpectNear, h->d_small + 5);
    h->guard = nullptr;
    if (rc) return rc;
    if (int rc2 = launch_envelope(h, 3, nullptr, nullptr, d_out_sdf, nullptr, nx, nys, nz, resolution, add_virtual_border,
                                  h->d_small, h->d_small + 5, s, 0, nullptr, &ex)) return rc2;
    if (!h->defer_fold) if (int rc3 = fold_slots(h, d_maxdsq, s)) return rc3;
    HIP_TRY(h, hipMemsetAsync(h->d_small, 0, 128, s));
    h->small_clean = true;
    return SDFGPU_OK;
}
int sdfgpu_sweep_x_lines_device(sdfgpu_handle h, const int32_t* d_plane_dsq, int64_t nx, int64_t nys, int64_t nz,
                                int64_t y_global, int64_t ny_global, double resolution, int add_virtual_border,
                                float* d_out_sdf, uint32_t* d_maxdsq, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_sweep_x_lines_device_body(h, d_plane_dsq, nx, nys, nz, y_global, ny_global, resolution, add_virtual_border, d_out_sdf, d_maxdsq, stream); });
}

static int sdfgpu_sweep_x_device_body(sdfgpu_handle h, const int32_t* d_plane_dsq, int64_t halo_lo, int64_t nxs,
                          int64_t halo_hi, int64_t ny, int64_t nz, int lo_truncated, int hi_truncated,
                          int64_t x_global, int64_t nx_global, double resolution, int add_virtual_border,
                          float* d_out_sdf, uint32_t* d_maxdsq, uint32_t* d_status, void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!d_plane_dsq || !d_out_sdf || !d_maxdsq)
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null device pointer");
    if (halo_lo < 0 || halo_hi < 0 || x_global < 0 || x_global + nxs > nx_global)
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "inconsistent slab geometry");
    h->guard = nullptr;
    h->far_y = nullptr;
    if (int rc = check_dims(h, halo_lo + nxs + halo_hi, ny, nz)) return rc;
    if (int rc = check_dims(h, nx_global, ny, nz)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = launch_sweep_x(h, d_plane_dsq, d_out_sdf, halo_lo, nxs, halo_hi, ny, nz, lo_truncated, hi_truncated,
                                x_global, nx_global, resolution, add_virtual_border, d_maxdsq, d_status,
                                (hipStream_t)stream)) return rc;
    return h->defer_fold ? SDFGPU_OK : fold_slots(h, d_maxdsq, (hipStream_t)stream);
}
int sdfgpu_sweep_x_device(sdfgpu_handle h, const int32_t* d_plane_dsq, int64_t halo_lo, int64_t nxs,
                          int64_t halo_hi, int64_t ny, int64_t nz, int lo_truncated, int hi_truncated,
                          int64_t x_global, int64_t nx_global, double resolution, int add_virtual_border,
                          float* d_out_sdf, uint32_t* d_maxdsq, uint32_t* d_status, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_sweep_x_device_body(h, d_plane_dsq, halo_lo, nxs, halo_hi, ny, nz, lo_truncated, hi_truncated, x_global, nx_global, resolution, add_virtual_border, d_out_sdf, d_maxdsq, d_status, stream); });
}

static int sdfgpu_pack_bits_device_body(sdfgpu_handle h, const uint8_t* d_filled, int64_t n_rows, int64_t nz, uint32_t* d_bits,
                            void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!d_filled || !d_bits) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null device pointer");
    if (n_rows <= 0 || nz <= 0 || (nz % 32) != 0) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "pack_bits needs nz % 32 == 0");
    HIP_TRY(h, hipSetDevice(h->device));
    return launch_pack_bits(h, d_filled, nullptr, 0, 0, 0, n_rows * nz, d_bits, (hipStream_t)stream);
}
int sdfgpu_pack_bits_device(sdfgpu_handle h, const uint8_t* d_filled, int64_t n_rows, int64_t nz, uint32_t* d_bits,
                            void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_pack_bits_device_body(h, d_filled, n_rows, nz, d_bits, stream); });
}

static int sdfgpu_dense_ball_device_body(sdfgpu_handle h, const uint32_t* d_bits, int64_t rows_x, int64_t out_lo, int64_t out_hi,
                             int64_t ny, int64_t nz, double resolution, float* d_out_sdf, uint32_t* d_maxdsq,
                             uint32_t* d_uncertified, void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!d_bits || !d_out_sdf || !d_maxdsq || !d_uncertified) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null device pointer");
    if (out_lo < 0 || out_hi <= out_lo || out_hi > rows_x) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "inconsistent slab geometry");
    if (int rc = check_dims(h, rows_x, ny, nz)) return rc;
    if (!dense_eligible(h, nz, 0)) return fail(h, SDFGPU_ERR_UNSUPPORTED_SIZE, "dense kernel needs nz = 32 * 2^k <= 2048");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = launch_ball_dense(h, d_bits, d_out_sdf, rows_x, out_lo, out_hi, ny, nz, resolution, d_maxdsq,
                                   d_uncertified, (hipStream_t)stream)) return rc;
    return h->defer_fold ? SDFGPU_OK : fold_slots(h, d_maxdsq, (hipStream_t)stream);
}
int sdfgpu_dense_ball_device(sdfgpu_handle h, const uint32_t* d_bits, int64_t rows_x, int64_t out_lo, int64_t out_hi,
                             int64_t ny, int64_t nz, double resolution, float* d_out_sdf, uint32_t* d_maxdsq,
                             uint32_t* d_uncertified, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_dense_ball_device_body(h, d_bits, rows_x, out_lo, out_hi, ny, nz, resolution, d_out_sdf, d_maxdsq, d_uncertified, stream); });
}

// One x slab of the dense path in three host calls instead of ten (the per-call cost of the Python binding is what
// limits a rank at ~0.15 ms per build): phase 0 = clear the status words + pack the boundary planes (the caller then
// posts the halo exchange), phase 1 = pack the interior + ball kernel on every plane that needs no neighbour data
// (10 / 11 = only the first / second half of that), phase 2 = ball kernel on the border planes + fold of the maxima
// (after the exchange has been waited for).  Without neighbours, or for slabs too thin to split, phase 0 packs
// everything, phase 1 does nothing and phase 2 runs the ball kernel on the whole slab.
static int sdfgpu_slab_dense_phase_body(sdfgpu_handle h, int phase, const uint8_t* d_mask_slab, int64_t nxs, int64_t ny, int64_t nz,
                            uint32_t* d_bits_ext, int64_t halo_lo, int64_t halo_hi, double resolution, float* d_out,
                            uint32_t* d_small, void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!d_mask_slab || !d_bits_ext || !d_out || !d_small) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null device pointer");
    const int64_t hb = kBallR;
    if (nxs <= 0 || (halo_lo != 0 && halo_lo != hb) || (halo_hi != 0 && halo_hi != hb))
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "inconsistent slab geometry");
    if (int rc = check_dims(h, halo_lo + nxs + halo_hi, ny, nz)) return rc;
    if (!dense_eligible(h, nz, 0)) return fail(h, SDFGPU_ERR_UNSUPPORTED_SIZE, "dense kernel needs nz = 32 * 2^k <= 2048");
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int64_t plane = ny * nz, wplane = ny * (nz / 32), rows_x = halo_lo + nxs + halo_hi;
    uint32_t* own = d_bits_ext + halo_lo * wplane;
    const bool split = (halo_lo || halo_hi) && 4 * hb < nxs;
    const int64_t i_lo = (split && halo_lo) ? hb : 0, i_hi = (split && halo_hi) ? nxs - hb : nxs;
    auto pack = [&](int64_t p0, int64_t p1) -> int {
        if (p1 <= p0) return SDFGPU_OK;
        return launch_pack_bits(h, d_mask_slab + p0 * plane, nullptr, 0, 0, 0, (p1 - p0) * plane, own + p0 * wplane, s);
    };
    auto ball = [&](int64_t p0, int64_t p1) -> int {
        if (p1 <= p0) return SDFGPU_OK;
        return launch_ball_dense(h, d_bits_ext, d_out + p0 * plane, rows_x, halo_lo + p0, halo_lo + p1, ny, nz, resolution,
                                 d_small, d_small + 3, s);
    };
    switch (phase) {
    case 0:
        HIP_TRY(h, hipMemsetAsync(d_small, 0, 16, s));
        if (!split) return pack(0, nxs);
        if (int rc = pack(0, hb)) return rc;
        return pack(nxs - hb, nxs);
    case 1:
    case 10:
    case 11:
        if (!split) return SDFGPU_OK;
        if (phase != 11) if (int rc = pack(hb, nxs - hb)) return rc;
        if (phase != 10) if (int rc = ball(i_lo, i_hi)) return rc;
        return SDFGPU_OK;
    case 2:
        if (!split) {
            if (int rc = ball(0, nxs)) return rc;
        } else {
            if (int rc = ball(0, i_lo)) return rc;
            if (int rc = ball(i_hi, nxs)) return rc;
        }
        return fold_slots(h, d_small, s);
    default:
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "unknown phase %d", phase);
    }
}
int sdfgpu_slab_dense_phase(sdfgpu_handle h, int phase, const uint8_t* d_mask_slab, int64_t nxs, int64_t ny, int64_t nz,
                            uint32_t* d_bits_ext, int64_t halo_lo, int64_t halo_hi, double resolution, float* d_out,
                            uint32_t* d_small, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_slab_dense_phase_body(h, phase, d_mask_slab, nxs, ny, nz, d_bits_ext, halo_lo, halo_hi, resolution, d_out, d_small, stream); });
}

static int sdfgpu_fold_extrema_device_body(sdfgpu_handle h, uint32_t* d_maxdsq, void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!d_maxdsq) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null device pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    return fold_slots(h, d_maxdsq, (hipStream_t)stream);
}
int sdfgpu_fold_extrema_device(sdfgpu_handle h, uint32_t* d_maxdsq, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_fold_extrema_device_body(h, d_maxdsq, stream); });
}

static int sdfgpu_build_tagged_cells_body(sdfgpu_handle h, const void* cells, size_t cell_stride, size_t occupancy_offset,
                              size_t object_id_offset, int object_mode, const uint32_t* object_ids, int64_t n_object_ids,
                              int unknown_is_filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                              int add_virtual_border, float* out_sdf, double* out_max, double* out_min) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!out_sdf) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null host pointer");
    if (cell_stride < 8 || (cell_stride % 4) || (occupancy_offset % 4) || (object_id_offset % 4) ||
        occupancy_offset + 4 > cell_stride || object_id_offset + 4 > cell_stride)
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "cell layout must be 4-byte aligned and in range");
    if (object_mode < 0 || object_mode > 2 || n_object_ids < 0 || n_object_ids > 0x7fffffffLL ||
        (object_mode == 2 && n_object_ids > 0 && !object_ids))
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "bad object filter");
    if (object_mode == 2 && n_object_ids == 0) object_mode = 0;       // "no objects supplied" = any object (:826)
    if (int rc = check_dims(h, nx, ny, nz)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const int64_t n = nx * ny * nz;
    if (!cells) {                                  // the records of the previous call on this handle, still on the device
        if (h->tag_cached_bytes == 0 || h->tag_cached_bytes != (size_t)n * cell_stride)
            return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "cells is null and the handle holds no cell records of this size");
    } else {
        h->tag_cached_bytes = 0;
        if (int rc = ensure(h, h->stage_in, (size_t)n * cell_stride, "input staging")) return rc;
    }
    if (int rc = ensure(h, h->stage_out, (size_t)n * 4, "output staging")) return rc;
    if (int rc = ensure(h, h->tagmask, (size_t)n, "tagged-object mask")) return rc;
    if (int rc = ensure(h, h->tagids, (size_t)std::max<int64_t>(n_object_ids, 1) * 4, "object id list")) return rc;
    if (cells) {
        if (int rc0 = copy_from_host(h, h->stage_in.ptr, cells, (size_t)n * cell_stride)) return rc0;
        h->tag_cached_bytes = (size_t)n * cell_stride;
    }
    if (n_object_ids > 0) {                        // sorted copy: the classify kernel binary-searches it
        std::vector<uint32_t> sorted_ids(object_ids, object_ids + n_object_ids);
        std::sort(sorted_ids.begin(), sorted_ids.end());
        HIP_TRY(h, hipMemcpy(h->tagids.ptr, sorted_ids.data(), (size_t)n_object_ids * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_classify_tagged, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr,
                       (const char*)h->stage_in.ptr, (int64_t)cell_stride, (int64_t)occupancy_offset, (int64_t)object_id_offset,
                       unknown_is_filled, object_mode, (const uint32_t*)h->tagids.ptr, (int)n_object_ids, n,
                       (uint8_t*)h->tagmask.ptr);
    HIP_TRY(h, hipGetLastError());
    int rc = build_device_impl(h, (const uint8_t*)h->tagmask.ptr, nullptr, 0, 0, 0, nx, ny, nz, resolution, add_virtual_border,
                               (float*)h->stage_out.ptr, nullptr);
    if (rc) return rc;
    if (int rc2 = copy_to_host(h, out_sdf, h->stage_out.ptr, (size_t)n * 4)) return rc2;
    double mx, mn;
    rc = sdfgpu_get_extrema(h, &mx, &mn);
    if (rc) return rc;
    if (out_max) *out_max = mx;
    if (out_min) *out_min = mn;
    return SDFGPU_OK;
}
int sdfgpu_build_tagged_cells(sdfgpu_handle h, const void* cells, size_t cell_stride, size_t occupancy_offset,
                              size_t object_id_offset, int object_mode, const uint32_t* object_ids, int64_t n_object_ids,
                              int unknown_is_filled, int64_t nx, int64_t ny, int64_t nz, double resolution,
                              int add_virtual_border, float* out_sdf, double* out_max, double* out_min) {
    return rz_wrap(h, nullptr, [&]() -> int { return sdfgpu_build_tagged_cells_body(h, cells, cell_stride, occupancy_offset, object_id_offset, object_mode, object_ids, n_object_ids, unknown_is_filled, nx, ny, nz, resolution, add_virtual_border, out_sdf, out_max, out_min); });
}

static int sdfgpu_classify_cells_device_body(sdfgpu_handle h, const void* d_cells, size_t cell_stride, size_t occupancy_offset,
                                 int unknown_is_filled, int64_t n_cells, uint8_t* d_mask, void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!d_cells || !d_mask || n_cells < 0) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "bad classify arguments");
    if (cell_stride < 4 || (cell_stride % 4) || (occupancy_offset % 4) || occupancy_offset + 4 > cell_stride)
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "cell_stride/occupancy_offset must be 4-byte aligned and in range");
    HIP_TRY(h, hipSetDevice(h->device));
    if (n_cells == 0) return SDFGPU_OK;
    CellLoader ld{reinterpret_cast<const char*>(d_cells), (int64_t)cell_stride, (int64_t)occupancy_offset, unknown_is_filled};
    hipLaunchKernelGGL(k_classify_cells, dim3((unsigned)((n_cells + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       ld, n_cells, d_mask);
    HIP_TRY(h, hipGetLastError());
    return SDFGPU_OK;
}
int sdfgpu_classify_cells_device(sdfgpu_handle h, const void* d_cells, size_t cell_stride, size_t occupancy_offset,
                                 int unknown_is_filled, int64_t n_cells, uint8_t* d_mask, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_classify_cells_device_body(h, d_cells, cell_stride, occupancy_offset, unknown_is_filled, n_cells, d_mask, stream); });
}

static int sdfgpu_voxelize_points_device_body(sdfgpu_handle h, const float* d_points, int64_t n_points, const double* origin,
                                  double resolution, int64_t nx, int64_t ny, int64_t nz, uint8_t* d_mask, int clear_first,
                                  void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!d_mask || !origin || (n_points > 0 && !d_points) || n_points < 0 || !(resolution > 0.0))
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "bad voxelize arguments");
    if (int rc = check_dims(h, nx, ny, nz)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (clear_first) HIP_TRY(h, hipMemsetAsync(d_mask, 0, (size_t)(nx * ny * nz), s));
    if (n_points > 0) {
        hipLaunchKernelGGL(k_voxelize_points, dim3((unsigned)((n_points + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d_points,
                           n_points, origin[0], origin[1], origin[2], resolution, nx, ny, nz, d_mask);
        HIP_TRY(h, hipGetLastError());
    }
    return SDFGPU_OK;
}
int sdfgpu_voxelize_points_device(sdfgpu_handle h, const float* d_points, int64_t n_points, const double* origin,
                                  double resolution, int64_t nx, int64_t ny, int64_t nz, uint8_t* d_mask, int clear_first,
                                  void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_voxelize_points_device_body(h, d_points, n_points, origin, resolution, nx, ny, nz, d_mask, clear_first, stream); });
}

static int sdfgpu_voxelize_points_bits_device_body(sdfgpu_handle h, const float* d_points, int64_t n_points, const double* origin,
                                       double resolution, int64_t nx, int64_t ny, int64_t nz, uint32_t* d_bits, int clear_first,
                                       void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!d_bits || !origin || (n_points > 0 && !d_points) || n_points < 0 || !(resolution > 0.0))
        return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "bad voxelize arguments");
    if (int rc = check_dims(h, nx, ny, nz)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (clear_first) HIP_TRY(h, hipMemsetAsync(d_bits, 0, (((size_t)(nx * ny * nz) + 31) / 32) * 4, s));
    if (n_points > 0) {
        hipLaunchKernelGGL(k_voxelize_points_bits, dim3((unsigned)((n_points + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, d_points,
                           n_points, origin[0], origin[1], origin[2], resolution, nx, ny, nz, d_bits);
        HIP_TRY(h, hipGetLastError());
    }
    return SDFGPU_OK;
}
int sdfgpu_voxelize_points_bits_device(sdfgpu_handle h, const float* d_points, int64_t n_points, const double* origin,
                                       double resolution, int64_t nx, int64_t ny, int64_t nz, uint32_t* d_bits, int clear_first,
                                       void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_voxelize_points_bits_device_body(h, d_points, n_points, origin, resolution, nx, ny, nz, d_bits, clear_first, stream); });
}

static int sdfgpu_gradient_device_body(sdfgpu_handle h, const float* d_sdf, int64_t nx, int64_t ny, int64_t nz,
                           double resolution, int enable_edge_gradients, void* d_out_grad, int out_is_f64,
                           void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!d_sdf || !d_out_grad) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null device pointer");
    if (int rc = check_dims(h, nx, ny, nz)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const int64_t n = nx * ny * nz;
    dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    // the reference's reciprocals, computed once with its own double operations (sdf.hpp:447 and :464-512)
    GradScale sc{};
    sc.inv2 = 1.0 / (2.0 * resolution);
    sc.inv_w1 = 1.0 / ((double)1 * resolution);
    sc.inv_w2 = 1.0 / ((double)2 * resolution);
    sc.inv2f = (float)sc.inv2;
    if (out_is_f64)
        hipLaunchKernelGGL(k_gradient<double>, grid, block, 0, s, d_sdf, (double*)d_out_grad, nx, ny, nz, sc, enable_edge_gradients);
    else if ((nz % 4) == 0 && ny * (nz / 4) < ((int64_t)1 << 31) - kBlock &&
             (reinterpret_cast<uintptr_t>(d_sdf) % 16) == 0 && (reinterpret_cast<uintptr_t>(d_out_grad) % 16) == 0) {
        const bool f32scale = (double)sc.inv2f == sc.inv2 && std::isfinite(sc.inv2) && std::fabs(sc.inv2) < 1e30 && std::fabs(sc.inv2) > 1e-30;
        const int64_t per_plane = ny * (nz / 4);                                  // groups of 4 voxels per x plane
        const dim3 g4((unsigned)((per_plane + kBlock - 1) / kBlock), (unsigned)std::min<int64_t>(nx, 65535));
        int gshift = -1;                                                           // log2(nz / 4) when that is a power of two
        for (int b = 0; b < 31; ++b) if (((int64_t)1 << b) == nz / 4) gshift = b;
        if (f32scale) hipLaunchKernelGGL(k_gradient_f32x4<true>, g4, block, 0, s, d_sdf, (float*)d_out_grad, nx, ny, nz, sc, enable_edge_gradients, gshift);
        else hipLaunchKernelGGL(k_gradient_f32x4<false>, g4, block, 0, s, d_sdf, (float*)d_out_grad, nx, ny, nz, sc, enable_edge_gradients, gshift);
    } else
        hipLaunchKernelGGL(k_gradient<float>, grid, block, 0, s, d_sdf, (float*)d_out_grad, nx, ny, nz, sc, enable_edge_gradients);
    HIP_TRY(h, hipGetLastError());
    return SDFGPU_OK;
}
int sdfgpu_gradient_device(sdfgpu_handle h, const float* d_sdf, int64_t nx, int64_t ny, int64_t nz,
                           double resolution, int enable_edge_gradients, void* d_out_grad, int out_is_f64,
                           void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_gradient_device_body(h, d_sdf, nx, ny, nz, resolution, enable_edge_gradients, d_out_grad, out_is_f64, stream); });
}

// Host-buffer form of the full-grid gradient (what SignedDistanceField::GetFullGradient / pysdf_tools'
// GetFullGradientNumpy call instead of N host GetGradient calls): upload the field, one kernel, download.
static int sdfgpu_gradient_body(sdfgpu_handle h, const float* sdf, int64_t nx, int64_t ny, int64_t nz, double resolution,
                    int enable_edge_gradients, void* out_grad, int out_is_f64) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!sdf || !out_grad) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null host pointer");
    if (int rc = check_dims(h, nx, ny, nz)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t n = (size_t)(nx * ny * nz), esz = out_is_f64 ? 8 : 4;
    // the x axis is cut into chunks (+1 plane of context each side) so that the staging stays <= ~1.5 GiB even
    // for the 24 B/voxel double output of a 512^3 field
    const int64_t plane = ny * nz;
    const int64_t max_rows = std::max<int64_t>(1, ((int64_t)1 << 26) / std::max<int64_t>(plane, 1));   // <= 64 Mi voxels per chunk
    const int64_t rows = std::min<int64_t>(nx, max_rows);
    h->tag_cached_bytes = 0;
    if (int rc = ensure(h, h->stage_in, (size_t)(rows + 2) * plane * 4, "input staging")) return rc;
    if (int rc = ensure(h, h->stage_out, (size_t)(rows + 2) * plane * 3 * esz, "output staging")) return rc;
    (void)n;
    for (int64_t x0 = 0; x0 < nx; x0 += rows) {
        const int64_t x1 = std::min(nx, x0 + rows);
        const int64_t lo = std::max<int64_t>(0, x0 - 1), hi = std::min(nx, x1 + 1);     // context planes
        if (int rc0 = copy_from_host(h, h->stage_in.ptr, sdf + lo * plane, (size_t)(hi - lo) * plane * 4)) return rc0;
        // the kernel treats the chunk as a grid of its own: its first / last plane would take the one-sided boundary
        // formula, so those planes are computed only when they ARE grid faces and are otherwise context that is skipped
        if (int rc = sdfgpu_gradient_device(h, (const float*)h->stage_in.ptr, hi - lo, ny, nz, resolution, enable_edge_gradients,
                                            h->stage_out.ptr, out_is_f64, nullptr)) return rc;
        if (int rc = copy_to_host(h, (char*)out_grad + (size_t)x0 * plane * 3 * esz,
                                  (const char*)h->stage_out.ptr + (size_t)(x0 - lo) * plane * 3 * esz,
                                  (size_t)(x1 - x0) * plane * 3 * esz)) return rc;
    }
    return SDFGPU_OK;
}
int sdfgpu_gradient(sdfgpu_handle h, const float* sdf, int64_t nx, int64_t ny, int64_t nz, double resolution,
                    int enable_edge_gradients, void* out_grad, int out_is_f64) {
    return rz_wrap(h, nullptr, [&]() -> int { return sdfgpu_gradient_body(h, sdf, nx, ny, nz, resolution, enable_edge_gradients, out_grad, out_is_f64); });
}

static int sdfgpu_query_points_device_body(sdfgpu_handle h, const float* d_sdf, int64_t nx, int64_t ny, int64_t nz, double resolution,
                               const double* world_to_grid, const double* grid_to_world_rotation, float oob_value,
                               const double* d_points, int64_t n_points, int enable_edge_gradients, double* d_distance,
                               double* d_gradient, uint8_t* d_flags, void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!d_sdf || (n_points > 0 && !d_points)) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null device pointer");
    if (n_points < 0 || !(resolution > 0.0)) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "bad point count or resolution");
    if (int rc = check_dims(h, nx, ny, nz)) return rc;
    if (n_points == 0 || (!d_distance && !d_gradient && !d_flags)) return SDFGPU_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    QueryArgs a{};
    a.sdf = d_sdf; a.points = d_points; a.distance = d_distance; a.gradient = d_gradient; a.flags = d_flags;
    a.n = n_points; a.nx = nx; a.ny = ny; a.nz = nz;
    a.res = resolution; a.inv_res = 1.0 / resolution; a.oob = (double)oob_value;
    static const double ident34[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    static const double ident33[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 12; ++i) a.w2g[i] = world_to_grid ? world_to_grid[i] : ident34[i];
    for (int i = 0; i < 9; ++i) a.rot[i] = grid_to_world_rotation ? grid_to_world_rotation[i] : ident33[i];
    a.edge = enable_edge_gradients;
    hipLaunchKernelGGL(k_query_points, dim3((unsigned)((n_points + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, a);
    HIP_TRY(h, hipGetLastError());
    return SDFGPU_OK;
}
int sdfgpu_query_points_device(sdfgpu_handle h, const float* d_sdf, int64_t nx, int64_t ny, int64_t nz, double resolution,
                               const double* world_to_grid, const double* grid_to_world_rotation, float oob_value,
                               const double* d_points, int64_t n_points, int enable_edge_gradients, double* d_distance,
                               double* d_gradient, uint8_t* d_flags, void* stream) {
    return rz_wrap(h, stream, [&]() -> int { return sdfgpu_query_points_device_body(h, d_sdf, nx, ny, nz, resolution, world_to_grid, grid_to_world_rotation, oob_value, d_points, n_points, enable_edge_gradients, d_distance, d_gradient, d_flags, stream); });
}

int sdfgpu_device_malloc(sdfgpu_handle h, size_t bytes, void** out_ptr) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!out_ptr) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "out_ptr is null");
    *out_ptr = nullptr;
    HIP_TRY(h, hipSetDevice(h->device));
    return rz_malloc(h, "sdfgpu_device_malloc", h->redzone ? std::max<size_t>(bytes, 4) : std::max<size_t>(bytes, 256), out_ptr);
}

int sdfgpu_device_free(sdfgpu_handle h, void* ptr) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!ptr) return SDFGPU_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->redzone) if (int rc = redzone_check(h, nullptr)) { (void)rz_free(h, ptr); return rc; }      // (last look at this buffer's zones)
    return rz_free(h, ptr);
}

static int sdfgpu_query_points_body(sdfgpu_handle h, const float* d_sdf, int64_t nx, int64_t ny, int64_t nz, double resolution,
                        const double* world_to_grid, const double* grid_to_world_rotation, float oob_value,
                        const double* points, int64_t n_points, int enable_edge_gradients, double* out_distance,
                        double* out_gradient, uint8_t* out_flags) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!d_sdf || (n_points > 0 && !points)) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "null pointer");
    if (n_points < 0 || !(resolution > 0.0)) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "bad point count or resolution");
    if (int rc = check_dims(h, nx, ny, nz)) return rc;
    if (n_points == 0 || (!out_distance && !out_gradient && !out_flags)) return SDFGPU_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    // staging in the context (grown on demand, re-used by the next call): points | distance | gradient | flags
    const size_t n = (size_t)n_points;
    const size_t o_d = n * 24, o_g = o_d + n * 8, o_f = o_g + n * 24, total = o_f + ((n + 255) & ~(size_t)255);
    if (int rc = ensure(h, h->query_stage, total, "query staging")) return rc;
    char* base = (char*)h->query_stage.ptr;
    // Round 5: the null stream of the context's device, ordered behind this handle's last build by its event --
    // not h->last_stream, which is a handle of the CALLER's (an earlier *_device build's stream may have been destroyed since).
    // A field produced ELSEWHERE (another handle, a caller's kernel) is ordered by the null stream's implicit synchronisation only
    // when its stream is a blocking one: producers on hipStreamNonBlocking streams (PyTorch's are) must be synchronised by the caller
    // before this call -- include/sdfgpu.h says so.
    hipStream_t s = nullptr;
    if (h->order_valid && h->order_stream != nullptr) HIP_TRY(h, hipStreamWaitEvent(s, h->build_done_ev, 0));
    if (int rc = copy_from_host(h, base, points, n * 24, s)) return rc;
    if (int rc = sdfgpu_query_points_device(h, d_sdf, nx, ny, nz, resolution, world_to_grid, grid_to_world_rotation, oob_value,
                                            (const double*)base, n_points, enable_edge_gradients,
                                            out_distance ? (double*)(base + o_d) : nullptr,
                                            out_gradient ? (double*)(base + o_g) : nullptr,
                                            out_flags ? (uint8_t*)(base + o_f) : nullptr, s)) return rc;
    if (out_distance) if (int rc = copy_to_host(h, out_distance, base + o_d, n * 8, s)) return rc;
    if (out_gradient) if (int rc = copy_to_host(h, out_gradient, base + o_g, n * 24, s)) return rc;
    if (out_flags) if (int rc = copy_to_host(h, out_flags, base + o_f, n, s)) return rc;
    return SDFGPU_OK;
}
int sdfgpu_query_points(sdfgpu_handle h, const float* d_sdf, int64_t nx, int64_t ny, int64_t nz, double resolution,
                        const double* world_to_grid, const double* grid_to_world_rotation, float oob_value,
                        const double* points, int64_t n_points, int enable_edge_gradients, double* out_distance,
                        double* out_gradient, uint8_t* out_flags) {
    return rz_wrap(h, nullptr, [&]() -> int { return sdfgpu_query_points_body(h, d_sdf, nx, ny, nz, resolution, world_to_grid, grid_to_world_rotation, oob_value, points, n_points, enable_edge_gradients, out_distance, out_gradient, out_flags); });
}

static int sdfgpu_debug_finish_table_body(sdfgpu_handle h, float* d_out, int64_t n, double resolution, int fast, uint32_t* out_slow_lanes) {
    if (!h || !d_out || n <= 0 || n > (1ll << 24)) return SDFGPU_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    if (int rc = ensure(h, h->tagids, 4, "object id list")) return rc;
    h->tag_cached_bytes = 0;
    HIP_TRY(h, hipMemset(h->tagids.ptr, 0, 4));
    const FinishFast fin = make_finish_fast(resolution, (uint64_t)n, fast != 0);
    hipLaunchKernelGGL(k_finish_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d_out, n, resolution, fin, (uint32_t*)h->tagids.ptr);
    HIP_TRY(h, hipGetLastError());
    uint32_t c = 0;
    HIP_TRY(h, hipMemcpy(&c, h->tagids.ptr, 4, hipMemcpyDeviceToHost));
    if (out_slow_lanes) *out_slow_lanes = c;
    return SDFGPU_OK;
}
int sdfgpu_debug_finish_table(sdfgpu_handle h, float* d_out, int64_t n, double resolution, int fast, uint32_t* out_slow_lanes) {
    return rz_wrap(h, nullptr, [&]() -> int { return sdfgpu_debug_finish_table_body(h, d_out, n, resolution, fast, out_slow_lanes); });
}

int sdfgpu_redzone_check(sdfgpu_handle h, void* stream) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    return redzone_check(h, (hipStream_t)stream);
}

int sdfgpu_debug_copy_zsweep(sdfgpu_handle h, int16_t* out_host, int64_t n) {
    if (!h || !out_host) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!h->have_result || n > h->last_n) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "no matching build");
    if (h->last_fused) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "the last build fused the z sweep into the y sweep: no z field exists");
    if (h->last_standby) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "the last build's stand-by pair takes its z distances from the bit field: no z field exists");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->order_valid) HIP_TRY(h, hipEventSynchronize(h->build_done_ev));
    HIP_TRY(h, hipMemcpy(out_host, h->zfield.ptr, (size_t)n * 2, hipMemcpyDeviceToHost));
    if (h->last_plane_skip && h->planebits.ptr && n == h->last_dims[0] * h->last_dims[1] * h->last_dims[2]) {
        // the z sweep of a build that went straight to the far-field pair does not write rows without a filled voxel
        const int64_t nrows = h->last_dims[0] * h->last_dims[1], nz = h->last_dims[2];
        std::vector<uint8_t> ra((size_t)nrows);
        HIP_TRY(h, hipMemcpy(ra.data(), h->planebits.ptr, ra.size(), hipMemcpyDeviceToHost));
        for (int64_t r = 0; r < nrows; ++r)
            if (!ra[(size_t)r]) std::fill(out_host + r * nz, out_host + (r + 1) * nz, (int16_t)kInf16);
    }
    return SDFGPU_OK;
}

int sdfgpu_debug_copy_yzsweep(sdfgpu_handle h, int32_t* out_host, int64_t n) {
    if (!h || !out_host) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!h->have_result || n > h->last_n) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "no matching build");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->order_valid) HIP_TRY(h, hipEventSynchronize(h->build_done_ev));
    HIP_TRY(h, hipMemcpy(out_host, h->yzfield.ptr, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (h->last_plane_skip && h->planebits.ptr && n == h->last_dims[0] * h->last_dims[1] * h->last_dims[2]) {
        // the y sweep skipped the x-planes without a filled voxel: every voxel of such a plane is "free, no filled voxel in the plane"
        std::vector<uint8_t> pa((size_t)h->last_dims[0]);
        HIP_TRY(h, hipMemcpy(pa.data(), (const uint8_t*)h->planebits.ptr + h->last_dims[0] * h->last_dims[1], pa.size(), hipMemcpyDeviceToHost));
        const int64_t plane = h->last_dims[1] * h->last_dims[2];
        for (int64_t x = 0; x < h->last_dims[0]; ++x)
            if (!pa[(size_t)x]) std::fill(out_host + x * plane, out_host + (x + 1) * plane, (int32_t)kInf32);
    }
    uint32_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};                 // status block of the last build ([7]: int32 hand-off used)
    HIP_TRY(h, hipMemcpy(st, h->d_result, sizeof st, hipMemcpyDeviceToHost));
    if (h->last_plane16 && st[7] == 0u && !h->last_predicted) {
        // 16-bit pipeline: the int32 buffer is the side table (valid only for saturated groups of 4)
        std::vector<int16_t> p16((size_t)n);
        HIP_TRY(h, hipMemcpy(p16.data(), h->plane16.ptr, (size_t)n * 2, hipMemcpyDeviceToHost));
        for (int64_t g = 0; g + 3 < n; g += 4) {
            bool sat = false;
            for (int k = 0; k < 4; ++k) sat |= std::abs((int)p16[(size_t)(g + k)]) >= 32767;
            if (!sat) for (int k = 0; k < 4; ++k) out_host[g + k] = p16[(size_t)(g + k)];
        }
    }
    return SDFGPU_OK;
}

int sdfgpu_set_profiling(sdfgpu_handle h, int enable) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    h->profiling = enable < 0 ? 0 : (enable > 3 ? 1 : enable);
    h->profiled_builds = 0;
    return SDFGPU_OK;
}

int sdfgpu_get_stage_times(sdfgpu_handle h, double* out_ms_sum, int64_t* out_builds) {
    if (!h || !out_ms_sum || !out_builds) return SDFGPU_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    for (int k = 0; k < 7; ++k) out_ms_sum[k] = 0.0;
    *out_builds = 0;
    if (h->events.empty()) return SDFGPU_OK;
    HIP_TRY(h, hipEventSynchronize(h->events.back()));
    for (size_t i = 0; i + 7 < h->events.size(); i += 8) {
        for (int k = 0; k < 7; ++k) {
            float ms = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&ms, h->events[i + k], h->events[i + k + 1]));
            out_ms_sum[k] += ms;
        }
        ++*out_builds;
    }
    for (size_t i = 0; i < h->events.size(); ++i)
        if (i % 8 == 0 || h->events[i] != h->events[i - 1]) h->event_pool.push_back(h->events[i]);   // marks may be shared
    h->events.clear();
    return SDFGPU_OK;
}

#ifdef SDFGPU_PHASE_CLOCKS
// profiling builds only: read (and clear) the phase clocks of the far-field kernels, [2][8] shader-clock sums over waves
extern "C" int sdfgpu_debug_read_clocks(sdfgpu_handle h, unsigned long long* out16) {
    if (!h || !out16) return SDFGPU_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    if (!h->d_clocks) return SDFGPU_OK;
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(out16, h->d_clocks, 16 * 8, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemset(h->d_clocks, 0, 16 * 8));
    return SDFGPU_OK;
}
#endif

int sdfgpu_set_option(sdfgpu_handle h, const char* name, int value) {
    if (!h || !name) return SDFGPU_ERR_INVALID_ARGUMENT;
    const std::string n(name);
    if (n == "fused_zy") { h->fused_zy = value != 0; h->fused_always = value == 2; }
    else if (n == "rows_per_chunk_y") h->tune_ty = value;
    else if (n == "rows_per_chunk_x") h->tune_tx = value;
    else if (n == "rows_per_chunk_zy") h->tune_tzy = value;
    else if (n == "fused_window") h->fused_h = value;
    else if (n == "plane16") h->plane16_on = value != 0;
    else if (n == "y16") h->y16_on = value != 0;
    else if (n == "probe_window") h->probe_window = value != 0;
    else if (n == "z_wave") h->z_wave_on = value != 0;
    else if (n == "dense") h->dense_on = value != 0;
    else if (n == "dense_generic") h->dense_generic_on = value != 0;
    else if (n == "envelope") h->envelope_on = value != 0;
    else if (n == "envelope_dc") h->envelope_dc = value != 0;
#ifdef SDFGPU_DEBUG_HOOKS
    else if (n == "dc_debug_stage") h->dc_debug_stage = value;
    else if (n == "dc_debug") h->dc_debug = value;              // profiling builds only: these skip work, results are then wrong
    else if (n == "ball_variant") h->ball_variant = value;
#endif
    else if (n == "i32_handoff") h->i32_handoff = value != 0;
    else if (n == "standby_far") h->standby_far = value != 0;
    else if (n == "host_pack") h->host_pack = (value >= 0 && value <= 2) ? value : 1;
    else if (n == "far_predict") h->far.set_mode(value);
    else if (n == "standby_fold") h->standby_fold = value != 0;
    else if (n == "standby_grid") h->standby_grid = value >= 32 ? value : 1024;
    else if (n == "expect_dense") h->pol.expect_dense = value != 0;      // tests: put the handle into the "dense tier trusted" state
    else if (n == "pack_variant") h->pack_variant = value;
    else if (n == "nt_store") h->nt_store = value;
    else if (n == "ball_block") h->ball_block = value;
    else if (n == "defer_fold") h->defer_fold = value != 0;
    else if (n == "policy_reset") { h->flags_pending = false; h->pol.reset(); h->far.reset(); h->flat_score_reset = true; }
    else if (n == "fixup") { h->pol.fixup_on = value != 0; h->pol.fix_mode = false; h->pol.dense3_mode = false; }
    else if (n == "dense3") { h->pol.dense3_on = value != 0; h->pol.dense3_mode = false; }
    else if (n == "dense3_mode") h->pol.dense3_mode = value != 0;
    else if (n == "dense_shell") h->shell_on = value != 0;
    else if (n == "dc_fixed") h->dc_fixed = value != 0;
    else if (n == "plane_skip") h->plane_skip = value != 0;
    else if (n == "flat_tiles") h->flat_tiles = value < 0 ? 0 : value > 2 ? 2 : value;
    else if (n == "redzone") {
        // from now on: every buffer the context holds is released (and comes back with -- or without -- zones when it is next needed);
        // the status block and the slots are replaced at once
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipDeviceSynchronize());
        for (DeviceBuffer* b : {&h->zfield, &h->yzfield, &h->plane16, &h->bits, &h->unc, &h->tileflag, &h->fix_order, &h->tagmask, &h->tagids, &h->stage_in,
                                &h->stage_bits, &h->stage_out, &h->query_stage, &h->planebits})
            if (b->ptr) { (void)rz_free(h, b->ptr); b->ptr = nullptr; b->bytes = 0; }
        h->tag_cached_bytes = 0;
        (void)rz_free(h, h->d_small);
        (void)rz_free(h, h->d_slots);
        h->d_small = h->d_slots = nullptr;
        h->redzone = value != 0;
        if (int rc = rz_malloc(h, "extrema slots", (size_t)kSlots * kSlotWords * 4, (void**)&h->d_slots)) return rc;
        if (int rc = rz_malloc(h, "status block", 512, (void**)&h->d_small)) return rc;
        HIP_TRY(h, hipMemset(h->d_slots, 0, (size_t)kSlots * kSlotWords * 4));
        HIP_TRY(h, hipMemset(h->d_small, 0, 512));
        h->d_result = h->d_small + 64;
        h->small_clean = true;
        h->have_result = false;
        h->flags_pending = h->far_pending = false;
    }
    else if (n == "dense3_fixed") h->dense3_fixed = value != 0;
    else if (n == "shell_min_words") h->shell_min_words = value >= 0 ? value : kShellMinWords;
    else if (n == "shell_budget_den") h->shell_budget_den = value >= 1 ? value : 8;
    else if (n == "dense3_staged") h->pol.dense3_staged = value != 0;
    else if (n == "fixup_mode") h->pol.fix_mode = value != 0;
    else if (n == "dense_retry") { h->pol.dense_retry = value; h->pol.dense_skip = 0; h->pol.dense_backoff = 0; }
    else if (n == "envelope_mode") { h->flags_pending = false; h->force_env = value != 0 ? 1 : -1; }
    else if (n == "far_threshold_y") h->far_thr[0] = value;
    else if (n == "far_threshold_x") h->far_thr[1] = value;
    else if (n == "far_fraction_den_y") h->far_den[0] = value > 0 ? value : 8;
    else if (n == "far_fraction_den_x") h->far_den[1] = value > 0 ? value : 24;
    else if (n == "x16_voxels_per_lane") h->x16_v = value;
    else if (n == "x16_window") h->x16_h = value;
    else if (n == "march_window") h->march_h = value == 8 ? 8 : 3;
    else if (n == "mid_threshold_y") h->mid_thr_y = value;
    else if (n == "mid_fraction_den_y") h->mid_den_y = value;
    else return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "unknown option '%s'", name);
    return SDFGPU_OK;
}

int sdfgpu_last_build_info(sdfgpu_handle h, int* out_fused_zy) {
    if (!h || !out_fused_zy) return SDFGPU_ERR_INVALID_ARGUMENT;
    *out_fused_zy = (h->last_fused ? 1 : 0) | (h->last_plane16 ? 2 : 0) | (h->last_dense ? 4 : 0) | (h->last_standby ? 8 : 0) |
                    (h->last_dense3 ? 16 : 0) | (h->last_staged ? 32 : 0) | (h->last_predicted ? 64 : 0);
    return SDFGPU_OK;
}

int sdfgpu_last_dense_certified(sdfgpu_handle h, int* out_certified) {
    if (!h || !out_certified) return SDFGPU_ERR_INVALID_ARGUMENT;
    if (!h->have_result) return fail(h, SDFGPU_ERR_INVALID_ARGUMENT, "no build has been issued on this handle");
    HIP_TRY(h, hipSetDevice(h->device));
    uint32_t v[8];
    // (the build's own event, not its stream: the stream is the caller's and may be gone by now -- ADVICE r4)
    if (h->order_valid) HIP_TRY(h, hipEventSynchronize(h->build_done_ev));
    HIP_TRY(h, hipMemcpy(v, h->d_result, sizeof v, hipMemcpyDeviceToHost));
    // bit 0: dense kernel decided everything; bit 1 / 2: the y / x sweep was redone by the envelope kernel
    uint32_t why = 0;
    HIP_TRY(h, hipMemcpy(&why, h->d_result + 21, sizeof why, hipMemcpyDeviceToHost));
    *out_certified = ((h->last_dense && v[3] == 0) ? 1 : 0) | (v[4] ? 2 : 0) | (v[5] ? 4 : 0) | (int)((why & 0xffu) << 8);
    return SDFGPU_OK;
}

int sdfgpu_debug_flat_habit(sdfgpu_handle h, int* out_score, int* out_gate) {
    if (!h || !out_score || !out_gate) return SDFGPU_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->order_valid) HIP_TRY(h, hipEventSynchronize(h->build_done_ev));
    int v[2] = {0, 0};
    HIP_TRY(h, hipMemcpy(v, h->d_small + 40, sizeof v, hipMemcpyDeviceToHost));
    *out_score = v[0];
    *out_gate = v[1];
    return SDFGPU_OK;
}

int sdfgpu_set_tuning(sdfgpu_handle h, int rows_per_chunk_y, int rows_per_chunk_x) {
    if (!h) return SDFGPU_ERR_INVALID_ARGUMENT;
    h->tune_ty = rows_per_chunk_y;
    h->tune_tx = rows_per_chunk_x;
    return SDFGPU_OK;
}

}  // extern "C"
