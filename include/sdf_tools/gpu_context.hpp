// gpu_context -- the per-thread libsdfgpu handle shared by the mirror headers (sdf_generation.hpp for the build,
// sdf.hpp for the full-grid gradient).  One GPU context per host thread (shared with the device-resident fields built on it): the C ABI is re-entrant
// per context (SURVEY.md 8(b) "Threading").  There is no CPU fallback: without a HIP device Get() throws std::runtime_error.
#pragma once
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>

#include "sdfgpu.h"
#ifdef SDF_TOOLS_MULTI_GPU
#include "sdfgpu_multi.h"       // link libsdfgpu_multi.so as well
#endif

namespace sdf_generation {

// The handle and what keeps it alive.  The context is created by (and normally used on) one host thread, but a result that
// stays in HBM (sdf_tools::DeviceSignedDistanceField) holds device memory of this context and may outlive its thread or be
// handed to another one (a worker thread that builds and returns the field; a pybind object collected elsewhere): the field
// therefore SHARES ownership -- the context is destroyed when the thread AND every field built on it are gone -- and every
// call on the handle is made under `mutex`, because the C ABI is re-entrant per context, not within one.
struct SharedGpuContext {
    sdfgpu_handle handle = nullptr;
    std::mutex mutex;
    SharedGpuContext() = default;
    SharedGpuContext(const SharedGpuContext&) = delete;
    SharedGpuContext& operator=(const SharedGpuContext&) = delete;
    ~SharedGpuContext() { if (handle) sdfgpu_destroy(handle); }
};

class GpuContext {
public:
    // this thread's context (created on first use; throws std::runtime_error without a HIP device)
    static std::shared_ptr<SharedGpuContext> Shared() {
        thread_local std::shared_ptr<SharedGpuContext> ctx;
        if (!ctx) {
            std::shared_ptr<SharedGpuContext> fresh = std::make_shared<SharedGpuContext>();
            const int rc = sdfgpu_create(DeviceIndex(), &fresh->handle);
            if (rc != SDFGPU_OK) throw std::runtime_error(std::string("sdfgpu: ") + sdfgpu_last_error(nullptr));
            ctx = std::move(fresh);
        }
        return ctx;
    }
    static sdfgpu_handle Get() { return Shared()->handle; }
    static int& DeviceIndex() { static int device = 0; return device; }
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
