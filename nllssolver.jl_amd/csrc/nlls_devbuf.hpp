// nlls_devbuf.hpp -- owning device buffer
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

namespace nlls {

// ---- device memory ---------------------------------------------------------------------------
// `owned == false`: the memory belongs to the context's hot arena (compact_hot_set, nlls_structure.cpp) -- release() then only forgets it.
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool owned = true;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), owned(o.owned) { o.p = nullptr; o.n = 0; o.owned = true; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; owned = o.owned; o.p = nullptr; o.n = 0; o.owned = true; } return *this; }
    ~DevBuf() { release(); }
    void release() { if (p && owned) (void)hipFree(p); p = nullptr; n = 0; owned = true; }
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
