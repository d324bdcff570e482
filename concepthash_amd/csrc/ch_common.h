// Shared device/host helpers for libconcepthash_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>

typedef uint16_t bf16_t;  // raw bf16 bits

using bf16x8 = __attribute__((ext_vector_type(8))) short;   // MFMA A/B fragment (4 VGPRs)
using bf16x4 = __attribute__((ext_vector_type(4))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;    // 16x16 accumulator
using f32x16 = __attribute__((ext_vector_type(16))) float;  // 32x32 accumulator

// round-to-nearest-even fp32 -> bf16 (plain cast compiles to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN)
__device__ __forceinline__ bf16_t f2bf(float x) {
    __bf16 b = (__bf16)x;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t x) { return __builtin_bit_cast(float, (uint32_t)x << 16); }
// two fp32 -> packed bf16x2 (lo in bits 0..15) with ONE v_cvt_pk_bf16_f32 (the scalar-cast form costs 2 cvt + shift + or)
typedef __bf16 ch_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float ch_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const ch_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ch_bf16x2_t));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- host-side error plumbing (ch_set_error, CH_REQUIRE: ch_host.h, shared with the plain C++ sources) -------
#include "ch_host.h"
#define CH_CHECK_HIP(expr)                                                                                   \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) {                                                                              \
            ch_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                                 \
            return 1;                                                                                        \
        }                                                                                                    \
    } while (0)
#define CH_LAUNCH_CHECK()                                                                                    \
    do {                                                                                                     \
        hipError_t _e = hipGetLastError();                                                                   \
        if (_e != hipSuccess) {                                                                              \
            ch_set_error(std::string("kernel launch: ") + hipGetErrorString(_e));                            \
            return 3;                                                                                        \
        }                                                                                                    \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: set it once per (kernel, device), thread-safely
// (two racing threads both set it -- idempotent).  One CH_LDS_ONCE object per kernel instantiation.
struct ch_once_per_device {
    std::atomic<uint64_t> done[4] = {};
};
static inline int ch_func_max_lds(const void *fn, int bytes, ch_once_per_device &once) {
    int dev = 0;
    CH_CHECK_HIP(hipGetDevice(&dev));
    std::atomic<uint64_t> &word = once.done[(dev >> 6) & 3];
    const uint64_t bit = 1ull << (dev & 63);
    if (word.load(std::memory_order_acquire) & bit) return 0;
    CH_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    word.fetch_or(bit, std::memory_order_release);
    return 0;
}

// ---- launch profiler hand-off (model.hip's mark() -> the NEXT kernel launch of this thread) ------------------------------------------
// When a pair is pending, CH_LAUNCH makes the launch with hipExtLaunchKernelGGL, so that (start, stop) carry the dispatch's own
// begin / end timestamps: the kernel's duration WITHOUT the launch boundary -- the quantity rocprofv3's kernel trace reports, so
// bench.py's per-kernel times and the tracked rocprofv3 summaries describe the same thing.  (An event recorded between two
// launches instead measures kernel + boundary: +3..5 us per launch, 8 % of a 52 us launch.)  Without a pending pair: a plain launch.
#include <hip/hip_ext.h>
struct ch_prof_pair {
    hipEvent_t start = nullptr, stop = nullptr;
    bool *used = nullptr;
};
extern thread_local ch_prof_pair g_ch_prof_pair;
#define CH_LAUNCH(kernel, grid, block, lds, stream, ...)                                                      \
    do {                                                                                                      \
        const ch_prof_pair _pp = g_ch_prof_pair;                                                              \
        g_ch_prof_pair = ch_prof_pair();                                                                      \
        if (_pp.start) {                                                                                      \
            *_pp.used = true;                                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, _pp.start, _pp.stop, 0, __VA_ARGS__);     \
        } else {                                                                                              \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                \
        }                                                                                                     \
    } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up64(int64_t a, int64_t b) { return ceil_div64(a, b) * b; }
