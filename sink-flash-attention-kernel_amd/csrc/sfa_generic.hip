// Generic exact-f32 sink attention kernels for gfx950.
//
// These kernels take any element type (f32 / f16 / bf16), any head dim up to 512
// and any N, and do all arithmetic in f32 on the vector ALU.  They are the path
// for fp32 tensors (gfx950 has no reduced-precision f32 MFMA mode and the
// reference's fp32 tests need true f32 products: tests/test_s_aux.py:198-238,
// tests/test_inference.py:54-91) and for head dims without an MFMA kernel.
// One wavefront owns one query row (forward, dQ) or one key row (dK/dV) and walks
// the two key ranges of the reference (sink range, then window range:
// sink_flash_attention.py:151-180) in chunks of 64 with lanes over keys for the
// score and lanes over the head dim for the value product.
//
// Also here: the backward preprocess (Delta = rowsum(dO*O), ds_aux partials) that
// replaces the eager ops at sink_flash_attention.py:582 and :658-665; the MFMA
// backward uses it too.
#include "sfa_common.hpp"
#include "sfa_internal.hpp"

namespace sfa {

namespace {

constexpr int kWaves = 4;

template <typename T, int VEC>
__device__ __forceinline__ float dot_row(const float* __restrict__ a, const T* __restrict__ row, int D) {
    float acc = 0.f;
    if constexpr (VEC == 4) {
        for (int d = 0; d < D; d += 4) {
            float x0, x1, x2, x3;
            if constexpr (sizeof(T) == 4) {
                const float4 r = *reinterpret_cast<const float4*>(row + d);
                x0 = r.x; x1 = r.y; x2 = r.z; x3 = r.w;
            } else {
                const uint2 r = *reinterpret_cast<const uint2*>(row + d);
                x0 = raw16_to_f32<T>((unsigned short)(r.x & 0xffff));
                x1 = raw16_to_f32<T>((unsigned short)(r.x >> 16));
                x2 = raw16_to_f32<T>((unsigned short)(r.y & 0xffff));
                x3 = raw16_to_f32<T>((unsigned short)(r.y >> 16));
            }
            acc = fmaf(a[d], x0, acc);
            acc = fmaf(a[d + 1], x1, acc);
            acc = fmaf(a[d + 2], x2, acc);
            acc = fmaf(a[d + 3], x3, acc);
        }
    } else {
        for (int d = 0; d < D; ++d) acc = fmaf(a[d], to_f32(row[d]), acc);
    }
    return acc;
}

__device__ __forceinline__ void row_ranges(int i, int num_sink, int window, int64_t beg[2], int64_t end[2]) {
    // range 0: sink keys the row can see; range 1: window keys not already in range 0
    beg[0] = 0;
    end[0] = num_sink < i + 1 ? num_sink : i + 1;
    if (end[0] < 0) end[0] = 0;
    int64_t ws = (int64_t)i - (int64_t)window + 1;
    if (ws < num_sink) ws = num_sink;
    if (ws < 0) ws = 0;
    beg[1] = ws;
    end[1] = (int64_t)i + 1;
}

// ------------------------------------------------------------------ forward
template <typename T, int DSLOTS, int VEC>
__global__ __launch_bounds__(kWaves * 64) void fwd_generic_kernel(View q, View k, View v, View o,
                                                                 float* __restrict__ lse,
                                                                 const float* __restrict__ s_aux, Problem p) {
    __shared__ float qs[kWaves][DSLOTS * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWaves + wave;
    const int h = blockIdx.y, b = blockIdx.z;
    const bool active = i < p.N;
    const int hk = h / (p.Hq / p.Hkv);
    const int D = p.D;
    if (active) {
        const T* qrow = reinterpret_cast<const T*>(q.ptr) + b * q.sb + h * q.sh + (int64_t)i * q.sn;
        for (int d = lane; d < D; d += 64) qs[wave][d] = to_f32(qrow[d]);
    }
    __syncthreads();
    if (!active) return;

    const T* kb = reinterpret_cast<const T*>(k.ptr) + b * k.sb + hk * k.sh;
    const T* vb = reinterpret_cast<const T*>(v.ptr) + b * v.sb + hk * v.sh;
    float m = s_aux ? s_aux[h] : -INFINITY;
    float l = s_aux ? 1.f : 0.f;
    float acc[DSLOTS];
#pragma unroll
    for (int t = 0; t < DSLOTS; ++t) acc[t] = 0.f;

    int64_t beg[2], end[2];
    row_ranges(i, p.num_sink, p.window, beg, end);
    for (int r = 0; r < 2; ++r) {
        for (int64_t c0 = beg[r]; c0 < end[r]; c0 += 64) {
            const int64_t j = c0 + lane;
            const bool valid = j < end[r];
            float s = -INFINITY;
            if (valid) s = dot_row<T, VEC>(qs[wave], kb + j * k.sn, D) * p.scale;
            const float cmax = wave_max(s);
            const float m_new = fmaxf(m, cmax);
            const float alpha = (m == -INFINITY) ? 0.f : expf(m - m_new);
            const float pj = valid ? expf(s - m_new) : 0.f;
            l = l * alpha + wave_sum(pj);
#pragma unroll
            for (int t = 0; t < DSLOTS; ++t) acc[t] *= alpha;
            const int cnt = (int)((end[r] - c0) < 64 ? (end[r] - c0) : 64);
            for (int jj = 0; jj < cnt; ++jj) {
                const float pp = __shfl(pj, jj, 64);
                const T* vr = vb + (c0 + jj) * v.sn;
#pragma unroll
                for (int t = 0; t < DSLOTS; ++t) {
                    const int d = lane + 64 * t;
                    if (d < D) acc[t] = fmaf(pp, to_f32(vr[d]), acc[t]);
                }
            }
            m = m_new;
        }
    }
    // l == 0 -> 1 (sink_flash_attention.py:183); LSE = m + log(l) (:192)
    if (l == 0.f) l = 1.f;
    T* orow = reinterpret_cast<T*>(o.ptr) + b * o.sb + h * o.sh + (int64_t)i * o.sn;
#pragma unroll
    for (int t = 0; t < DSLOTS; ++t) {
        const int d = lane + 64 * t;
        if (d < D) orow[d] = from_f32<T>(acc[t] / l);
    }
    if (lane == 0) lse[((int64_t)b * p.Hq + h) * p.N + i] = m + logf(l);
}

// ------------------------------------------------------- backward preprocess
// delta[b,h,i] = sum_d dO*O ; dsaux_part[(b*Hq+h)*nblk + blk] = -sum_rows exp(s_aux[h]-lse)*delta
constexpr int kPreRows = 64;  // rows per block (16 per wave)

template <typename T>
__global__ __launch_bounds__(256) void bwd_preprocess_kernel(View o, View d_o, const float* __restrict__ lse,
                                                            const float* __restrict__ s_aux,
                                                            float* __restrict__ delta,
                                                            float* __restrict__ dsaux_part, Problem p) {
    __shared__ float part[kWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = blockIdx.y, b = blockIdx.z;
    const int D = p.D;
    const T* ob = reinterpret_cast<const T*>(o.ptr) + b * o.sb + h * o.sh;
    const T* dob = reinterpret_cast<const T*>(d_o.ptr) + b * d_o.sb + h * d_o.sh;
    const int64_t rowbase = ((int64_t)b * p.Hq + h) * p.N;
    const float sa = s_aux ? s_aux[h] : 0.f;
    float wsum = 0.f;
    for (int r = 0; r < kPreRows / kWaves; ++r) {
        const int i = blockIdx.x * kPreRows + wave * (kPreRows / kWaves) + r;
        if (i >= p.N) break;
        const T* orow = ob + (int64_t)i * o.sn;
        const T* dorow = dob + (int64_t)i * d_o.sn;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s = fmaf(to_f32(orow[d]), to_f32(dorow[d]), s);
        s = wave_sum(s);
        if (lane == 0) delta[rowbase + i] = s;
        if (s_aux) wsum -= expf(sa - lse[rowbase + i]) * s;
    }
    if (s_aux) {
        if (lane == 0) part[wave] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int w = 0; w < kWaves; ++w) t += part[w];
            dsaux_part[((int64_t)b * p.Hq + h) * gridDim.x + blockIdx.x] = t;
        }
    }
}

// Same contract for 16-bit tensors with 16-byte aligned rows (every MFMA-path call): a row is spread over LPR lanes
// x 16 bytes, so one wave instruction fetches 64 / LPR whole rows (1 KiB, coalesced) and all of a wave's 16 rows of O
// and dO are in flight at once.  HBM-bound: reads O and dO once.  Optionally also emits the row constants of the
// dK/dV kernels (consts[b,h,0,:] = -LSE * lse_factor, consts[b,h,1,:] = -Delta) instead of a separate pass.
template <typename T, int LPR>
__global__ __launch_bounds__(256) void bwd_preprocess_vec_kernel(View o, View d_o, const float* __restrict__ lse,
                                                                const float* __restrict__ s_aux,
                                                                float* __restrict__ delta,
                                                                float* __restrict__ dsaux_part,
                                                                float* __restrict__ consts, Problem p, float lse_factor) {
    constexpr int RPI = 64 / LPR;                      // rows per wave instruction
    constexpr int NI = (kPreRows / kWaves) / RPI;      // instructions per wave (16 rows)
    __shared__ float part[kWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = blockIdx.y, b = blockIdx.z;
    const int chunk = lane % LPR, rsub = lane / LPR;
    const bool cact = chunk * 8 < p.D;
    const char* ob = o.ptr + ((int64_t)b * o.sb + (int64_t)h * o.sh) * 2 + (cact ? chunk * 16 : 0);
    const char* dob = d_o.ptr + ((int64_t)b * d_o.sb + (int64_t)h * d_o.sh) * 2 + (cact ? chunk * 16 : 0);
    const int64_t rowbase = ((int64_t)b * p.Hq + h) * p.N;
    const int row0 = blockIdx.x * kPreRows + wave * (kPreRows / kWaves) + rsub;
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    u4 xo[NI], xd[NI];
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        const int i = row0 + t * RPI;
        const int ic = i < p.N ? i : p.N - 1;
        xo[t] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(ob + (int64_t)ic * o.sn * 2));
        xd[t] = __builtin_nontemporal_load(reinterpret_cast<const u4*>(dob + (int64_t)ic * d_o.sn * 2));
    }
    const float sa = s_aux ? s_aux[h] : 0.f;
    float wsum = 0.f;
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        const int i = row0 + t * RPI;
        const unsigned ow[4] = {xo[t][0], xo[t][1], xo[t][2], xo[t][3]}, dw[4] = {xd[t][0], xd[t][1], xd[t][2], xd[t][3]};
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s = fmaf(raw16_to_f32<T>((unsigned short)(ow[e] & 0xffffu)), raw16_to_f32<T>((unsigned short)(dw[e] & 0xffffu)), s);
            s = fmaf(raw16_to_f32<T>((unsigned short)(ow[e] >> 16)), raw16_to_f32<T>((unsigned short)(dw[e] >> 16)), s);
        }
        if (!cact) s = 0.f;
#pragma unroll
        for (int off = 1; off < LPR; off <<= 1) s += __shfl_xor(s, off, 64);
        if (i < p.N && chunk == 0) {
            delta[rowbase + i] = s;
            const float ls = lse[rowbase + i];
            if (consts) {
                consts[rowbase * 2 + i] = -ls * lse_factor;
                consts[rowbase * 2 + p.N + i] = -s;
            }
            if (s_aux) wsum -= expf(sa - ls) * s;
        }
    }
    if (s_aux) {
        wsum = wave_sum(wsum);
        if (lane == 0) part[wave] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int w = 0; w < kWaves; ++w) t += part[w];
            dsaux_part[((int64_t)b * p.Hq + h) * gridDim.x + blockIdx.x] = t;
        }
    }
}

// ds_aux[h] = sum over (b, blk) of the partials, fixed order => deterministic
__global__ __launch_bounds__(256) void dsaux_reduce_kernel(const float* __restrict__ part, float* __restrict__ ds_aux,
                                                          int B, int Hq, int nblk) {
    __shared__ float red[256];
    const int h = blockIdx.x;
    float s = 0.f;
    const int total = B * nblk;
    for (int t = threadIdx.x; t < total; t += 256) {
        const int b = t / nblk, blk = t % nblk;
        s += part[((int64_t)b * Hq + h) * nblk + blk];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) ds_aux[h] = red[0];
}

// --------------------------------------------------------------- backward dQ
template <typename T, int DSLOTS, int VEC>
__global__ __launch_bounds__(kWaves * 64) void bwd_dq_generic_kernel(View q, View k, View v, View d_o, View dq,
                                                                    const float* __restrict__ lse,
                                                                    const float* __restrict__ delta, Problem p) {
    __shared__ float qs[kWaves][DSLOTS * 64];
    __shared__ float dos[kWaves][DSLOTS * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * kWaves + wave;
    const int h = blockIdx.y, b = blockIdx.z;
    const bool active = i < p.N;
    const int hk = h / (p.Hq / p.Hkv);
    const int D = p.D;
    if (active) {
        const T* qrow = reinterpret_cast<const T*>(q.ptr) + b * q.sb + h * q.sh + (int64_t)i * q.sn;
        const T* dorow = reinterpret_cast<const T*>(d_o.ptr) + b * d_o.sb + h * d_o.sh + (int64_t)i * d_o.sn;
        for (int d = lane; d < D; d += 64) {
            qs[wave][d] = to_f32(qrow[d]);
            dos[wave][d] = to_f32(dorow[d]);
        }
    }
    __syncthreads();
    if (!active) return;
    const T* kb = reinterpret_cast<const T*>(k.ptr) + b * k.sb + hk * k.sh;
    const T* vb = reinterpret_cast<const T*>(v.ptr) + b * v.sb + hk * v.sh;
    const int64_t ridx = ((int64_t)b * p.Hq + h) * p.N + i;
    const float lse_i = lse[ridx], delta_i = delta[ridx];
    float acc[DSLOTS];
#pragma unroll
    for (int t = 0; t < DSLOTS; ++t) acc[t] = 0.f;
    int64_t beg[2], end[2];
    row_ranges(i, p.num_sink, p.window, beg, end);
    for (int r = 0; r < 2; ++r) {
        for (int64_t c0 = beg[r]; c0 < end[r]; c0 += 64) {
            const int64_t j = c0 + lane;
            const bool valid = j < end[r];
            float ds = 0.f;
            if (valid) {
                const float s = dot_row<T, VEC>(qs[wave], kb + j * k.sn, D) * p.scale;
                const float dp = dot_row<T, VEC>(dos[wave], vb + j * v.sn, D);
                const float pj = expf(s - lse_i);
                ds = pj * (dp - delta_i);
            }
            const int cnt = (int)((end[r] - c0) < 64 ? (end[r] - c0) : 64);
            for (int jj = 0; jj < cnt; ++jj) {
                const float dd = __shfl(ds, jj, 64);
                const T* kr = kb + (c0 + jj) * k.sn;
#pragma unroll
                for (int t = 0; t < DSLOTS; ++t) {
                    const int d = lane + 64 * t;
                    if (d < D) acc[t] = fmaf(dd, to_f32(kr[d]), acc[t]);
                }
            }
        }
    }
    T* dqrow = reinterpret_cast<T*>(dq.ptr) + b * dq.sb + h * dq.sh + (int64_t)i * dq.sn;
#pragma unroll
    for (int t = 0; t < DSLOTS; ++t) {
        const int d = lane + 64 * t;
        if (d < D) dqrow[d] = from_f32<T>(acc[t] * p.scale);
    }
}

// ------------------------------------------------------------ backward dK/dV
// One wave per key row j of one KV head; the whole GQA group is summed in registers
// (the reference writes per-Q-head dK/dV and sums in PyTorch: sink_flash_attention.py:585-586,648-651).
template <typename T, int DSLOTS, int VEC>
__global__ __launch_bounds__(kWaves * 64) void bwd_dkdv_generic_kernel(View q, View k, View v, View d_o, View dk,
                                                                      View dv, const float* __restrict__ lse,
                                                                      const float* __restrict__ delta, Problem p) {
    __shared__ float ks[kWaves][DSLOTS * 64];
    __shared__ float vs[kWaves][DSLOTS * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * kWaves + wave;
    const int hk = blockIdx.y, b = blockIdx.z;
    const bool active = j < p.N;
    const int g = p.Hq / p.Hkv;
    const int D = p.D;
    if (active) {
        const T* krow = reinterpret_cast<const T*>(k.ptr) + b * k.sb + hk * k.sh + (int64_t)j * k.sn;
        const T* vrow = reinterpret_cast<const T*>(v.ptr) + b * v.sb + hk * v.sh + (int64_t)j * v.sn;
        for (int d = lane; d < D; d += 64) {
            ks[wave][d] = to_f32(krow[d]);
            vs[wave][d] = to_f32(vrow[d]);
        }
    }
    __syncthreads();
    if (!active) return;
    float acck[DSLOTS], accv[DSLOTS];
#pragma unroll
    for (int t = 0; t < DSLOTS; ++t) acck[t] = accv[t] = 0.f;
    // query rows that see key j: i in [j, N) for a sink key, else [j, min(N, j+W))
    const int64_t i_beg = j;
    int64_t i_end;
    if (j < p.num_sink) {
        i_end = p.N;
    } else {
        i_end = (int64_t)j + (int64_t)(p.window > 0 ? p.window : 0);
        if (i_end > p.N) i_end = p.N;
    }
    for (int hh = 0; hh < g; ++hh) {
        const int h = hk * g + hh;
        const T* qb = reinterpret_cast<const T*>(q.ptr) + b * q.sb + h * q.sh;
        const T* dob = reinterpret_cast<const T*>(d_o.ptr) + b * d_o.sb + h * d_o.sh;
        const float* lse_h = lse + ((int64_t)b * p.Hq + h) * p.N;
        const float* delta_h = delta + ((int64_t)b * p.Hq + h) * p.N;
        for (int64_t c0 = i_beg; c0 < i_end; c0 += 64) {
            const int64_t i = c0 + lane;
            const bool valid = i < i_end;
            float pj = 0.f, ds = 0.f;
            if (valid) {
                const float s = dot_row<T, VEC>(ks[wave], qb + i * q.sn, D) * p.scale;
                const float dp = dot_row<T, VEC>(vs[wave], dob + i * d_o.sn, D);
                pj = expf(s - lse_h[i]);
                ds = pj * (dp - delta_h[i]);
            }
            const int cnt = (int)((i_end - c0) < 64 ? (i_end - c0) : 64);
            for (int ii = 0; ii < cnt; ++ii) {
                const float pp = __shfl(pj, ii, 64);
                const float dd = __shfl(ds, ii, 64);
                const T* qr = qb + (c0 + ii) * q.sn;
                const T* dor = dob + (c0 + ii) * d_o.sn;
#pragma unroll
                for (int t = 0; t < DSLOTS; ++t) {
                    const int d = lane + 64 * t;
                    if (d < D) {
                        accv[t] = fmaf(pp, to_f32(dor[d]), accv[t]);
                        acck[t] = fmaf(dd, to_f32(qr[d]), acck[t]);
                    }
                }
            }
        }
    }
    T* dkrow = reinterpret_cast<T*>(dk.ptr) + b * dk.sb + hk * dk.sh + (int64_t)j * dk.sn;
    T* dvrow = reinterpret_cast<T*>(dv.ptr) + b * dv.sb + hk * dv.sh + (int64_t)j * dv.sn;
#pragma unroll
    for (int t = 0; t < DSLOTS; ++t) {
        const int d = lane + 64 * t;
        if (d < D) {
            dkrow[d] = from_f32<T>(acck[t] * p.scale);
            dvrow[d] = from_f32<T>(accv[t]);
        }
    }
}

// ----------------------------------------------------------------- dispatch
inline bool vec4_ok(const sfa_tensor* t) {
    const int es = dtype_size(t->dtype);
    const int64_t bytes = 4 * es;  // 4 elements per load
    return t->shape[3] % 4 == 0 && ((uintptr_t)t->ptr % bytes) == 0 && t->stride[0] % 4 == 0 &&
           t->stride[1] % 4 == 0 && t->stride[2] % 4 == 0;
}

#define SFA_DISPATCH_DSLOTS(D, ...)                       \
    [&] {                                                 \
        if ((D) <= 64) { constexpr int DSLOTS = 1; return __VA_ARGS__(); }  \
        if ((D) <= 128) { constexpr int DSLOTS = 2; return __VA_ARGS__(); } \
        if ((D) <= 256) { constexpr int DSLOTS = 4; return __VA_ARGS__(); } \
        constexpr int DSLOTS = 8;                         \
        return __VA_ARGS__();                             \
    }()

#define SFA_DISPATCH_DTYPE(dt, ...)                                        \
    [&] {                                                                  \
        if ((dt) == SFA_DTYPE_F32) { using T = float; return __VA_ARGS__(); }   \
        if ((dt) == SFA_DTYPE_F16) { using T = f16_t; return __VA_ARGS__(); }   \
        using T = bf16_t;                                                  \
        return __VA_ARGS__();                                              \
    }()

}  // namespace

int fwd_generic(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o, float* lse,
                const float* s_aux, const Problem& p, hipStream_t stream) {
    if (p.D > 512) {
        set_error("generic kernels support head dim <= 512, got %d", p.D);
        return SFA_ERR_UNSUPPORTED;
    }
    const bool vec = vec4_ok(q) && vec4_ok(k);
    dim3 grid((unsigned)cdiv64(p.N, kWaves), p.Hq, p.B), block(kWaves * 64);
    SFA_DISPATCH_DTYPE(q->dtype, [&] {
        SFA_DISPATCH_DSLOTS(p.D, [&] {
            if (vec)
                fwd_generic_kernel<T, DSLOTS, 4><<<grid, block, 0, stream>>>(make_view(q), make_view(k), make_view(v),
                                                                            make_view(o), lse, s_aux, p);
            else
                fwd_generic_kernel<T, DSLOTS, 1><<<grid, block, 0, stream>>>(make_view(q), make_view(k), make_view(v),
                                                                            make_view(o), lse, s_aux, p);
        });
    });
    set_path("fwd_generic_f32math");
    return launch_status("fwd_generic");
}

static bool rows16(const sfa_tensor* t) {
    return ((uintptr_t)t->ptr % 16) == 0 && (t->stride[0] * 2) % 16 == 0 && (t->stride[1] * 2) % 16 == 0 &&
           (t->stride[2] * 2) % 16 == 0;
}

bool bwd_preprocess_vectorised(const sfa_tensor* o, const sfa_tensor* d_o, const Problem& p) {
    return o->dtype != SFA_DTYPE_F32 && p.D % 8 == 0 && p.D <= 128 && rows16(o) && rows16(d_o);
}

int bwd_preprocess(const sfa_tensor* o, const sfa_tensor* d_o, const float* lse, const float* s_aux, float* delta,
                   float* dsaux_part, float* ds_aux, const Problem& p, hipStream_t stream, float* consts, float lse_factor) {
    const int nblk = (int)cdiv64(p.N, kPreRows);
    dim3 grid(nblk, p.Hq, p.B), block(256);
    const bool vec = bwd_preprocess_vectorised(o, d_o, p);
    if (vec) {
        const int lpr = p.D <= 32 ? 4 : (p.D <= 64 ? 8 : 16);
#define SFA_PRE(TT, L) bwd_preprocess_vec_kernel<TT, L><<<grid, block, 0, stream>>>(make_view(o), make_view(d_o), lse, s_aux, delta, dsaux_part, consts, p, lse_factor)
        if (o->dtype == SFA_DTYPE_BF16) {
            if (lpr == 4) SFA_PRE(bf16_t, 4); else if (lpr == 8) SFA_PRE(bf16_t, 8); else SFA_PRE(bf16_t, 16);
        } else {
            if (lpr == 4) SFA_PRE(f16_t, 4); else if (lpr == 8) SFA_PRE(f16_t, 8); else SFA_PRE(f16_t, 16);
        }
#undef SFA_PRE
    } else {
        if (consts) {
            set_error("bwd_preprocess: the row constants need the vectorised path");
            return SFA_ERR_UNSUPPORTED;
        }
        SFA_DISPATCH_DTYPE(o->dtype, [&] {
            bwd_preprocess_kernel<T><<<grid, block, 0, stream>>>(make_view(o), make_view(d_o), lse, s_aux, delta,
                                                                dsaux_part, p);
        });
    }
    int st = launch_status("bwd_preprocess");
    if (st != SFA_OK) return st;
    if (s_aux) {
        dsaux_reduce_kernel<<<dim3(p.Hq), dim3(256), 0, stream>>>(dsaux_part, ds_aux, p.B, p.Hq, nblk);
        st = launch_status("dsaux_reduce");
    }
    return st;
}

int64_t bwd_preprocess_nblk(int64_t N) { return cdiv64(N, kPreRows); }

int bwd_generic(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* d_o,
                const float* lse, const float* delta, const sfa_tensor* dq, const sfa_tensor* dk,
                const sfa_tensor* dv, const Problem& p, hipStream_t stream) {
    if (p.D > 512) {
        set_error("generic kernels support head dim <= 512, got %d", p.D);
        return SFA_ERR_UNSUPPORTED;
    }
    const bool vec = vec4_ok(q) && vec4_ok(k) && vec4_ok(v) && vec4_ok(d_o);
    dim3 block(kWaves * 64);
    dim3 grid_q((unsigned)cdiv64(p.N, kWaves), p.Hq, p.B);
    dim3 grid_k((unsigned)cdiv64(p.N, kWaves), p.Hkv, p.B);
    SFA_DISPATCH_DTYPE(q->dtype, [&] {
        SFA_DISPATCH_DSLOTS(p.D, [&] {
            if (vec) {
                bwd_dkdv_generic_kernel<T, DSLOTS, 4><<<grid_k, block, 0, stream>>>(
                    make_view(q), make_view(k), make_view(v), make_view(d_o), make_view(dk), make_view(dv), lse,
                    delta, p);
                record_stage(2, stream);
                bwd_dq_generic_kernel<T, DSLOTS, 4><<<grid_q, block, 0, stream>>>(
                    make_view(q), make_view(k), make_view(v), make_view(d_o), make_view(dq), lse, delta, p);
            } else {
                bwd_dkdv_generic_kernel<T, DSLOTS, 1><<<grid_k, block, 0, stream>>>(
                    make_view(q), make_view(k), make_view(v), make_view(d_o), make_view(dk), make_view(dv), lse,
                    delta, p);
                record_stage(2, stream);
                bwd_dq_generic_kernel<T, DSLOTS, 1><<<grid_q, block, 0, stream>>>(
                    make_view(q), make_view(k), make_view(v), make_view(d_o), make_view(dq), lse, delta, p);
            }
        });
    });
    set_path("bwd_generic_f32math");
    return launch_status("bwd_generic");
}

}  // namespace sfa
