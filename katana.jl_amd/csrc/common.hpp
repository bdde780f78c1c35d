// common.hpp -- error handling and device buffers for the Katana HIP engine.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <utility>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/katana_hip.h"

namespace ktn {

struct Error : public std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define KTN_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            throw ::ktn::Error(e__ == hipErrorOutOfMemory ? KTN_E_NOMEM : KTN_E_HIP,          \
                               std::string(#expr) + ": " + hipGetErrorString(e__) + " (" +    \
                                   __FILE__ + ":" + std::to_string(__LINE__) + ")");          \
        }                                                                                      \
    } while (0)

#define KTN_REQUIRE(cond, msg)                                   \
    do {                                                         \
        if (!(cond)) throw ::ktn::Error(KTN_E_INVALID, (msg));   \
    } while (0)

// Growable device array.  resize() keeps the first min(old,new) elements.
template <typename T>
struct DBuf {
    T* p = nullptr;
    size_t n = 0;    // logical size
    size_t cap = 0;  // allocated elements

    DBuf() = default;
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    ~DBuf() { release(); }

    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = cap = 0;
    }
    void reserve(size_t want, hipStream_t s) {
        if (want <= cap) return;
        size_t ncap = cap ? cap : 256;
        while (ncap < want) ncap += ncap / 2 + 256;
        T* q = nullptr;
        KTN_HIP(hipMalloc(&q, ncap * sizeof(T)));
        if (p && n) KTN_HIP(hipMemcpyAsync(q, p, n * sizeof(T), hipMemcpyDeviceToDevice, s));
        if (p) {
            KTN_HIP(hipStreamSynchronize(s));
            (void)hipFree(p);
        }
        p = q;
        cap = ncap;
    }
    void resize(size_t want, hipStream_t s) {
        reserve(want, s);
        n = want;
    }
    void swap(DBuf& o) {
        std::swap(p, o.p);
        std::swap(n, o.n);
        std::swap(cap, o.cap);
    }
    void zero(hipStream_t s) {
        if (n) KTN_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
    }
    void upload(const T* host, size_t count, hipStream_t s) {
        resize(count, s);
        if (count) KTN_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void upload(const std::vector<T>& v, hipStream_t s) { upload(v.data(), v.size(), s); }
    void download(T* host, size_t count, hipStream_t s) const {
        if (count) KTN_HIP(hipMemcpyAsync(host, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
        KTN_HIP(hipStreamSynchronize(s));
    }
    std::vector<T> to_host(hipStream_t s) const {
        std::vector<T> v(n);
        download(v.data(), n, s);
        return v;
    }
};

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace ktn
