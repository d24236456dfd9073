// gpu_context -- the per-thread libsdfgpu handle shared by the mirror headers (sdf_generation.hpp for the build,
// sdf.hpp for the full-grid gradient).  One GPU context per host thread: the C ABI is re-entrant per context
// (SURVEY.md 8(b) "Threading").  There is no CPU fallback: without a HIP device Get() throws std::runtime_error.
#pragma once
#include <stdexcept>
#include <string>

#include "sdfgpu.h"
#ifdef SDF_TOOLS_MULTI_GPU
#include "sdfgpu_multi.h"       // link libsdfgpu_multi.so as well
#endif

namespace sdf_generation {

class GpuContext {
public:
    static sdfgpu_handle Get() {
        thread_local GpuContext ctx;
        if (!ctx.handle_) {
            const int rc = sdfgpu_create(DeviceIndex(), &ctx.handle_);
            if (rc != SDFGPU_OK) throw std::runtime_error(std::string("sdfgpu: ") + sdfgpu_last_error(nullptr));
        }
        return ctx.handle_;
    }
    static int& DeviceIndex() { static int device = 0; return device; }
    ~GpuContext() { if (handle_) sdfgpu_destroy(handle_); }
private:
    sdfgpu_handle handle_ = nullptr;
};

#ifdef SDF_TOOLS_MULTI_GPU
// Several GPUs of one node behind the same seam (sdfgpu_init(n_gpus)): after SetNumGpus(n > 1) the
// ExtractSignedDistanceField overloads cut the grid into x slabs and build it with sdfgpu_multi_build (RCCL exchange
// between the slabs); results are bit-identical to one GPU.
class MultiGpuContext {
public:
    static int& NumGpus() { static int n = 1; return n; }
    static void SetNumGpus(const int n) { NumGpus() = n < 1 ? 1 : n; }
    static sdfgpu_multi_handle Get() {
        thread_local MultiGpuContext ctx;
        if (ctx.handle_ && ctx.ranks_ != NumGpus()) { sdfgpu_multi_destroy(ctx.handle_); ctx.handle_ = nullptr; }
        if (!ctx.handle_) {
            const int rc = sdfgpu_multi_create(NumGpus(), nullptr, &ctx.handle_);
            if (rc != SDFGPU_OK) throw std::runtime_error(std::string("sdfgpu_multi: ") + sdfgpu_multi_last_error(nullptr));
            ctx.ranks_ = NumGpus();
        }
        return ctx.handle_;
    }
    ~MultiGpuContext() { if (handle_) sdfgpu_multi_destroy(handle_); }
private:
    sdfgpu_multi_handle handle_ = nullptr;
    int ranks_ = 0;
};

inline void ThrowOnMultiStatus(sdfgpu_multi_handle h, const int rc) {
    if (rc == SDFGPU_OK) return;
    const std::string msg = std::string("sdfgpu_multi: ") + sdfgpu_multi_last_error(h);
    if (rc == SDFGPU_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}
#endif

inline void ThrowOnStatus(sdfgpu_handle h, const int rc) {
    if (rc == SDFGPU_OK) return;
    const std::string msg = std::string("sdfgpu: ") + sdfgpu_last_error(h);
    if (rc == SDFGPU_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

}  // namespace sdf_generation
