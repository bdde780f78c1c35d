// launch.hpp -- launch macros shared by the translation units that launch kernels (G lanes per output selected at run time).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "common.hpp"
#include "types.hpp"

#define LAUNCH_G(G, KERNEL, count, stream, ...)                                                          \
    do {                                                                                                 \
        const int64_t cnt__ = (count);                                                                   \
        if (cnt__ > 0) {                                                                                 \
            switch (G) {                                                                                 \
                case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(ceil_div(cnt__ * 4, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break;   \
                case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(ceil_div(cnt__ * 8, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break;   \
                case 16: hipLaunchKernelGGL((KERNEL<16>), dim3(ceil_div(cnt__ * 16, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
                case 32: hipLaunchKernelGGL((KERNEL<32>), dim3(ceil_div(cnt__ * 32, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
                default: hipLaunchKernelGGL((KERNEL<64>), dim3(ceil_div(cnt__ * 64, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            }                                                                                            \
        }                                                                                                \
    } while (0)

#define LAUNCH_GB(G, KERNEL, B, count, stream, ...)                                                      \
    do {                                                                                                 \
        const int64_t cnt__ = (count);                                                                   \
        if (cnt__ > 0) {                                                                                 \
            switch (G) {                                                                                 \
                case 4: hipLaunchKernelGGL((KERNEL<4, B>), dim3(ceil_div(cnt__ * 4, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break;   \
                case 8: hipLaunchKernelGGL((KERNEL<8, B>), dim3(ceil_div(cnt__ * 8, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break;   \
                case 16: hipLaunchKernelGGL((KERNEL<16, B>), dim3(ceil_div(cnt__ * 16, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
                case 32: hipLaunchKernelGGL((KERNEL<32, B>), dim3(ceil_div(cnt__ * 32, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
                default: hipLaunchKernelGGL((KERNEL<64, B>), dim3(ceil_div(cnt__ * 64, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); break; \
            }                                                                                            \
        }                                                                                                \
    } while (0)

// same, with the kernel's own start/stop timestamps recorded into (E0, E1)
#define LAUNCH_G_EV(G, KERNEL, count, stream, E0, E1, ...)                                               \
    do {                                                                                                 \
        const int64_t cnt__ = (count);                                                                   \
        if (cnt__ > 0) {                                                                                 \
            switch (G) {                                                                                 \
                case 4: hipExtLaunchKernelGGL((KERNEL<4>), dim3(ceil_div(cnt__ * 4, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break;   \
                case 8: hipExtLaunchKernelGGL((KERNEL<8>), dim3(ceil_div(cnt__ * 8, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break;   \
                case 16: hipExtLaunchKernelGGL((KERNEL<16>), dim3(ceil_div(cnt__ * 16, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
                case 32: hipExtLaunchKernelGGL((KERNEL<32>), dim3(ceil_div(cnt__ * 32, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
                default: hipExtLaunchKernelGGL((KERNEL<64>), dim3(ceil_div(cnt__ * 64, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
            }                                                                                            \
        }                                                                                                \
    } while (0)

#define LAUNCH_GB_EV(G, KERNEL, B, count, stream, E0, E1, ...)                                           \
    do {                                                                                                 \
        const int64_t cnt__ = (count);                                                                   \
        if (cnt__ > 0) {                                                                                 \
            switch (G) {                                                                                 \
                case 4: hipExtLaunchKernelGGL((KERNEL<4, B>), dim3(ceil_div(cnt__ * 4, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break;   \
                case 8: hipExtLaunchKernelGGL((KERNEL<8, B>), dim3(ceil_div(cnt__ * 8, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break;   \
                case 16: hipExtLaunchKernelGGL((KERNEL<16, B>), dim3(ceil_div(cnt__ * 16, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
                case 32: hipExtLaunchKernelGGL((KERNEL<32, B>), dim3(ceil_div(cnt__ * 32, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
                default: hipExtLaunchKernelGGL((KERNEL<64, B>), dim3(ceil_div(cnt__ * 64, kBlock)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
            }                                                                                            \
        }                                                                                                \
    } while (0)

// G lanes per output, T outputs per lane group: grid = ceil(count * G / (kBlock * T))
#define LAUNCH_GT(G, T, KERNEL, count, stream, E0, E1, ...)                                              \
    do {                                                                                                 \
        const int64_t cnt__ = (count);                                                                   \
        if (cnt__ > 0) {                                                                                 \
            switch (G) {                                                                                 \
                case 4: hipExtLaunchKernelGGL((KERNEL<4, T>), dim3(ceil_div(cnt__ * 4, kBlock * T)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break;   \
                case 8: hipExtLaunchKernelGGL((KERNEL<8, T>), dim3(ceil_div(cnt__ * 8, kBlock * T)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break;   \
                case 16: hipExtLaunchKernelGGL((KERNEL<16, T>), dim3(ceil_div(cnt__ * 16, kBlock * T)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
                case 32: hipExtLaunchKernelGGL((KERNEL<32, T>), dim3(ceil_div(cnt__ * 32, kBlock * T)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
                default: hipExtLaunchKernelGGL((KERNEL<64, T>), dim3(ceil_div(cnt__ * 64, kBlock * T)), dim3(kBlock), 0, stream, E0, E1, 0, __VA_ARGS__); break; \
            }                                                                                            \
        }                                                                                                \
    } while (0)

#define LAUNCH_1(KERNEL, count, stream, ...)                                                             \
    do {                                                                                                 \
        const int64_t cnt__ = (count);                                                                   \
        if (cnt__ > 0) hipLaunchKernelGGL(KERNEL, dim3(ceil_div(cnt__, kBlock)), dim3(kBlock), 0, stream, __VA_ARGS__); \
    } while (0)
