// Shared host/device helpers for libsfa (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "sfa.h"

namespace sfa {

// ---------------------------------------------------------------- host side
void set_error(const char* fmt, ...);
void set_path(const char* fmt, ...);
void record_stage(int i, hipStream_t stream);  // no-op unless sfa_debug_set_stage_events() armed it
bool stage_events_armed();
extern void* g_debug_ptr;                       // sfa_debug_set_ptr(): device buffer of the stamped diagnostic bodies
extern int g_variant[8];                        // sfa_debug_set_variant(): 0 = dK/dV body, 1 = forward, 2 = dQ, 3 = dK/dV workgroup order, 7 = tickets off
// A/B hook: read by -DSFA_AB development builds only; a release build compiles every such branch away
#ifdef SFA_AB
inline int variant(int which) { return __atomic_load_n(&g_variant[which], __ATOMIC_RELAXED); }
#else
inline int variant(int) { return 0; }
#endif

#define SFA_CHECK_ARG(cond, ...)             \
    do {                                     \
        if (!(cond)) {                       \
            ::sfa::set_error(__VA_ARGS__);   \
            return SFA_ERR_INVALID_ARGUMENT; \
        }                                    \
    } while (0)

// Dynamic LDS above the default limit needs hipFuncAttributeMaxDynamicSharedMemorySize, and function attributes are
// per DEVICE: a process that drives several GPUs (HF device_map="auto", pipeline stages) launches the same kernel on
// all of them.  `flags` is a per-call-site bitmask of the devices already prepared.
inline void ensure_dynamic_lds(const void* kernel, int bytes, unsigned long long* flags) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    // two threads may both set the (idempotent) attribute; the flag word itself is only ever OR-ed atomically
    if (!(__atomic_load_n(flags, __ATOMIC_ACQUIRE) & bit)) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        __atomic_fetch_or(flags, bit, __ATOMIC_RELEASE);
    }
}

// integer tuning knob: the default in a release build; -DSFA_AB development builds read SFA_<NAME> from the environment,
// ONCE per call site (getenv is not free and not thread-safe against setenv): static const int knob = env_int("SFA_X", default);
inline int env_int(const char* name, int dflt) {
#ifdef SFA_AB
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return SFA_OK;
}

// A [B,H,N,D] view with element strides (D stride == 1), passed to kernels by value.
struct View {
    char* ptr;
    int64_t sb, sh, sn;  // element strides
};

inline View make_view(const sfa_tensor* t) {
    return View{(char*)t->ptr, t->stride[0], t->stride[1], t->stride[2]};
}

// Problem description shared by the prefill kernels.  Packed (varlen) batches: cu != nullptr is a DEVICE array of
// n_seq + 1 row offsets into tensors of shape [1, H, n_total, D]; B = n_seq and N = the longest sequence then only
// size the grids, every workgroup takes its own (first row, length) from cu.
struct Problem {
    int B, Hq, Hkv, N, D;
    int num_sink, window;
    float scale;
    const int* cu = nullptr;
    int n_total = 0;
    int Nk = 0;          // key rows when they outnumber the query rows (0 = same as N): row i sits at position i + Nk - N
};

#ifdef __HIPCC__
struct SeqInfo {
    int N;      // rows of this workgroup's sequence
    int row0;   // its first row inside the (b, head) slice
    int bb;     // batch index for pointer arithmetic
};
__device__ __forceinline__ SeqInfo seq_of(const int* cu, int b, int N) {
    if (cu) {
        const int r0 = cu[b];
        return SeqInfo{cu[b + 1] - r0, r0, 0};
    }
    return SeqInfo{N, 0, b};
}
// byte extent of one sequence's rows of a (b, head) slice: the buffer descriptor range, so that rows >= N read zero
// and stores to them are dropped (a packed neighbour starts right there)
__device__ __forceinline__ unsigned seq_range(const int* cu, unsigned whole, int N, int64_t sn, int D) {
    return cu ? (unsigned)((((int64_t)N - 1) * sn + D) * 2) : whole;
}
#endif

inline int dtype_size(int dt) { return dt == SFA_DTYPE_F32 ? 4 : 2; }

inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- device side
using bf16_t = __hip_bfloat16;
using f16_t = __half;

template <typename T> struct DT;
template <> struct DT<float> { static constexpr int id = SFA_DTYPE_F32; };
template <> struct DT<f16_t> { static constexpr int id = SFA_DTYPE_F16; };
template <> struct DT<bf16_t> { static constexpr int id = SFA_DTYPE_BF16; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(f16_t x) { return __half2float(x); }
__device__ __forceinline__ float to_f32(bf16_t x) { return __bfloat162float(x); }

template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float x) { return __float2half_rn(x); }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return __float2bfloat16(x); }

// bf16 / f16 bit helpers on raw 16-bit payloads
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

template <typename T> __device__ __forceinline__ float raw16_to_f32(unsigned short b);
template <> __device__ __forceinline__ float raw16_to_f32<bf16_t>(unsigned short b) { return bf16_bits_to_f32(b); }
template <> __device__ __forceinline__ float raw16_to_f32<f16_t>(unsigned short b) {
    return __half2float(__ushort_as_half(b));
}

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

}  // namespace sfa
