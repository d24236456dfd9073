// gpu_context -- the per-thread libsdfgpu handle shared by the mirror headers (sdf_generation.hpp for the build,
// sdf.hpp for the full-grid gradient).  One GPU context per host thread: the C ABI is re-entrant per context
// (SURVEY.md 8(b) "Threading").  There is no CPU fallback: without a HIP device Get() throws std::runtime_error.
#pragma once
#include <stdexcept>
#include <string>

#include "sdfgpu.h"

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

inline void ThrowOnStatus(sdfgpu_handle h, const int rc) {
    if (rc == SDFGPU_OK) return;
    const std::string msg = std::string("sdfgpu: ") + sdfgpu_last_error(h);
    if (rc == SDFGPU_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

}  // namespace sdf_generation
