#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace ktn {
size_t scan_i64_temp_bytes(size_t n);
hipError_t exclusive_scan_i64(void* temp, size_t temp_bytes, const int64_t* in, int64_t* out,
                              size_t n, hipStream_t s);
size_t sort_pairs_temp_bytes(size_t n);
hipError_t sort_pairs_u64_u32(void* temp, size_t temp_bytes, const uint64_t* kin, uint64_t* kout,
                              const uint32_t* vin, uint32_t* vout, size_t n, int begin_bit, int end_bit,
                              hipStream_t s);
size_t sort_keys_desc_temp_bytes(size_t n);
hipError_t sort_keys_desc_u64(void* temp, size_t temp_bytes, const uint64_t* kin, uint64_t* kout, size_t n, hipStream_t s);
}  // namespace ktn
