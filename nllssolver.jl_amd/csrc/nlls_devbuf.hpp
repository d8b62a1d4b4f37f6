// nlls_devbuf.hpp -- owning device buffer
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

namespace nlls {

// ---- device memory ---------------------------------------------------------------------------
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~DevBuf() { release(); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    hipError_t alloc(size_t count) {
        release(); n = count;
        if (count == 0) { return hipSuccess; }
        return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    }
    hipError_t upload(const std::vector<T>& h) {
        hipError_t e = alloc(h.size()); if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

}  // namespace nlls
