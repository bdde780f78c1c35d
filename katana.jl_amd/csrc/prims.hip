// prims.hip -- device-wide scan and sort used to rebuild the LP's column mirror.
// rocPRIM (AMD's native primitives library) is used directly; these are setup-time
// utilities, not hot-path kernels.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include "prims.hpp"

namespace ktn {

size_t scan_i64_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan((void*)nullptr, bytes, (const int64_t*)nullptr, (int64_t*)nullptr,
                                  int64_t(0), n, rocprim::plus<int64_t>(), (hipStream_t)0);
    return bytes;
}

hipError_t exclusive_scan_i64(void* temp, size_t temp_bytes, const int64_t* in, int64_t* out,
                              size_t n, hipStream_t s) {
    return rocprim::exclusive_scan(temp, temp_bytes, in, out, int64_t(0), n,
                                   rocprim::plus<int64_t>(), s);
}

size_t sort_pairs_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs((void*)nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                    (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 64, (hipStream_t)0);
    return bytes;
}

hipError_t sort_pairs_u64_u32(void* temp, size_t temp_bytes, const uint64_t* kin, uint64_t* kout,
                              const uint32_t* vin, uint32_t* vout, size_t n, int begin_bit, int end_bit,
                              hipStream_t s) {
    // LSD radix sort: stable, so sorting on a bit range keeps the input order among equal keys
    return rocprim::radix_sort_pairs(temp, temp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, s);
}

size_t sort_keys_desc_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_keys_desc((void*)nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, n, 0, 64,
                                        (hipStream_t)0);
    return bytes;
}

hipError_t sort_keys_desc_u64(void* temp, size_t temp_bytes, const uint64_t* kin, uint64_t* kout, size_t n, hipStream_t s) {
    return rocprim::radix_sort_keys_desc(temp, temp_bytes, kin, kout, n, 0, 64, s);
}

}  // namespace ktn
