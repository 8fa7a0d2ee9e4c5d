// Single-query decode attention for gfx950: split-KV streaming kernel + reduce kernel.
//
// Replaces sink_attention/decode_kernel.py:28-113 (_decode_split_kv_kernel) and the
// ~10-op PyTorch phase 2 at :201-226 of the reference.  The path is HBM-bound
// (K and V are each read exactly once), so the design is about bytes in flight:
//
//   * one workgroup = (batch, KV head, KV split); ALL q heads of the GQA group are
//     served from the same K/V registers (the reference re-reads K/V per q head);
//   * a key row is spread over LPK lanes, 16 bytes per lane, so every wave load is a
//     fully coalesced 1 KiB nontemporal read (K/V are read once: keep them out of L2);
//   * each lane group keeps its own online-softmax state (m, l, acc) for U keys in
//     flight; partial dot products are reduced over the LPK lanes with DPP row ops;
//   * partial (m, l, o) per split go to the caller's workspace; the reduce kernel
//     folds them and the s_aux "virtual split" (m = s_aux, l = 1, o = 0:
//     decode_kernel.py:205-215) and writes q-dtype output.
#include "sfa_common.hpp"
#include "sfa_internal.hpp"

namespace sfa {

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

// sum over the LPK consecutive lanes that hold one key row; every lane gets the total
template <int LPK>
__device__ __forceinline__ float group_sum(float x) {
    if constexpr (LPK >= 2) x += dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
    if constexpr (LPK >= 4) x += dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
    if constexpr (LPK >= 8) x += dpp_mov<0x141>(x);  // row_half_mirror
    if constexpr (LPK >= 16) x += dpp_mov<0x140>(x); // row_mirror
    if constexpr (LPK >= 32) x += __shfl_xor(x, 16, 64);
    if constexpr (LPK >= 64) x += __shfl_xor(x, 32, 64);
    return x;
}

template <typename T, int EPL>
__device__ __forceinline__ void cvt16B(const u32x4& raw, float (&f)[EPL]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = __uint_as_float(raw[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[2 * e] = raw16_to_f32<T>((unsigned short)(raw[e] & 0xffffu));
            f[2 * e + 1] = raw16_to_f32<T>((unsigned short)(raw[e] >> 16));
        }
    }
}

// Keys come from up to two segments (k, v: rows [0, N1); k2, v2: rows [N1, Nkv)): a contiguous cache has N1 = Nkv;
// the sink + ring cache passes the sink buffer and the window ring (softmax is order-invariant, so the ring is read
// in place, no linearisation copy).
// Fold the split partials of one (b, h) and the s_aux virtual split (decode_kernel.py:205-226); `nthr` threads
// starting at `tid` = 0 cooperate (D <= 4 * nthr).
// COH (one-pass mode): the partials were written by workgroups on OTHER XCDs in this same launch.  They are stored and
// fetched as relaxed agent-scope atomics - `sc1` stores that write through the XCD's L2 and `sc1` loads that do not hit
// its non-coherent lines - so that no agent-scope FENCE (an L2 write-back + invalidate per workgroup) is needed around
// the arrival counter.
template <bool COH>
__device__ __forceinline__ float ld_part(const float* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
__device__ __forceinline__ void st_part(float* p, float v, bool coh) {
    if (coh) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// W = lanes that share the split statistics by shuffles (64: a whole wave per head; 32: a half-wave per head, two heads per
// wave, head dims up to 128).
template <typename T, bool COH = false, int W = 64>
__device__ __forceinline__ void reduce_head(const float* __restrict__ Mp, const float* __restrict__ Lp,
                                            const float* __restrict__ Op, const float* __restrict__ s_aux, const View& o,
                                            int Hq, int S, int D, int b, int h, int tid, int nthr) {
    // Latency-bound (in one-pass mode every load is a round trip to the memory side): the statistics of the first W splits
    // (lane = split) and the O rows of the first U splits are requested TOGETHER, the weights of those splits come from the
    // statistics registers by shuffles; further splits (S > W or S > U) take more rounds of U independent loads per thread.
    const int64_t base = ((int64_t)b * Hq + h) * S;
    const int wl = tid & (W - 1);
    constexpr int U = 8;
    const float sa = s_aux ? s_aux[h] : -INFINITY;
    const float m0 = wl < S ? ld_part<COH>(Mp + base + wl) : -INFINITY;
    const float l0 = wl < S ? ld_part<COH>(Lp + base + wl) : 0.f;
    float x[U][4];
    auto fetch = [&](int s0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = s0 + u < S ? s0 + u : S - 1;
            const float* op = Op + (base + s) * D;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int d = tid + nthr * t;
                x[u][t] = d < D ? ld_part<COH>(op + d) : 0.f;
            }
        }
    };
    fetch(0);
    float mloc = m0;
    for (int s0 = W; s0 < S; s0 += W) {
        const int s = s0 + wl;
        if (s < S) mloc = fmaxf(mloc, ld_part<COH>(Mp + base + s));
    }
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) mloc = fmaxf(mloc, __shfl_xor(mloc, off, 64));
    const float mstar = fmaxf(mloc, sa);
    float lloc = l0 * ((m0 == -INFINITY) ? 0.f : __expf(m0 - mstar));
    for (int s0 = W; s0 < S; s0 += W) {
        const int s = s0 + wl;
        if (s < S) {
            const float ms = ld_part<COH>(Mp + base + s);
            lloc += ld_part<COH>(Lp + base + s) * ((ms == -INFINITY) ? 0.f : __expf(ms - mstar));
        }
    }
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) lloc += __shfl_xor(lloc, off, 64);
    float L = lloc + ((sa == -INFINITY) ? 0.f : __expf(sa - mstar));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += U) {
        if (s0 > 0) fetch(s0);
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = s0 + u;
            float ms = -INFINITY;
            if (s < W) ms = __shfl(m0, s, W);                       // (lanes >= S hold -inf)
            else if (s < S) ms = ld_part<COH>(Mp + base + s);
            w[u] = (ms != -INFINITY) ? __expf(ms - mstar) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = fmaf(x[u][t], w[u], acc[t]);
    }
    L = fmaxf(L, 1e-8f);  // clamp(min=1e-8), decode_kernel.py:222
    T* orow = reinterpret_cast<T*>(o.ptr) + (int64_t)b * o.sb + (int64_t)h * o.sh;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int d = tid + nthr * t;
        if (d < D) orow[d] = from_f32<T>(acc[t] / L);
    }
}

// One-pass mode (SFA_FLAG_DECODE_ONE_PASS): `cnt` = zero-initialised int32 [B * Hkv + 1] counters that the kernel
// leaves zero.  The LAST split of a (b, KV head) to arrive (relaxed atomic counter; the partials travel as write-through
// stores / sc1 loads, see ld_part) folds the partials of that head group itself, so there is no second launch; the last
// workgroup of the whole grid advances the device state.
struct OnePass {
    int* cnt;
    const float* s_aux;
    View o;
};

// Fused cache step (sfa_decode_ring_step): `fr.slot` >= 0 names the ring slot the token being decoded goes to.  That
// slot's OLD content (the evicted token, or nothing yet) is never read: the key at that position comes from fr.kn /
// fr.vn, and the first split's first lanes store it into the ring, so append + attention are one launch.
// `dyn` (sfa_decode_ring_step_dyn): the cache state lives on the DEVICE as int32 {sink_len, window_len, write_pos}, so
// a whole generation step can sit in a captured hipGraph: the split kernel takes its key counts and the write slot
// from there (the grid is sized for the full cache; splits past the end produce empty partials) and the reduce kernel,
// which the stream orders after every reader, advances the state.
struct Fresh {
    View kn, vn;
    int slot;
    int* dyn;       // device {sink_len, window_len before the append, write_pos} or null
    int wsize;      // ring capacity (dyn only)
};

template <typename T, int LPK, int GT>
__global__ __launch_bounds__(256) void decode_split_kernel(View q, View k, View v, View k2, View v2, int N1, Fresh fr,
                                                          OnePass op1,
                                                          float* __restrict__ Mp, float* __restrict__ Lp,
                                                          float* __restrict__ Op, int Hq, int Hkv, int Nkv, int D,
                                                          int kps, int S, float scale) {
    constexpr int EPL = 16 / sizeof(T);
    constexpr int KPW = 64 / LPK;
    constexpr int NSTREAM = 4 * KPW;
    constexpr int U = (GT <= 4) ? 8 : 4;
    constexpr int DP = LPK * EPL;  // padded head dim covered by a lane group

    __shared__ float sm[4][GT];
    __shared__ float sl[4][GT];
    __shared__ float so[4][GT][DP];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = lane % LPK, kg = lane / LPK;
    const bool dact = chunk * EPL < D;
    const int d0 = dact ? chunk * EPL : 0;
    const int split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
    const int g = Hq / Hkv;
    if (fr.dyn) {
        N1 = fr.dyn[0];
        const int wl = fr.dyn[1] + 1 < fr.wsize ? fr.dyn[1] + 1 : fr.wsize;
        Nkv = N1 + wl;
        fr.slot = fr.dyn[2];
    }
    const int k_beg = split * kps;
    const int k_end = (k_beg + kps < Nkv) ? (k_beg + kps) : Nkv;
    const int sid = wave * KPW + kg;

    const T* kb = reinterpret_cast<const T*>(k.ptr) + (int64_t)b * k.sb + (int64_t)hk * k.sh + d0;
    const T* vb = reinterpret_cast<const T*>(v.ptr) + (int64_t)b * v.sb + (int64_t)hk * v.sh + d0;
    const T* kb2 = reinterpret_cast<const T*>(k2.ptr) + (int64_t)b * k2.sb + (int64_t)hk * k2.sh + d0;
    const T* vb2 = reinterpret_cast<const T*>(v2.ptr) + (int64_t)b * v2.sb + (int64_t)hk * v2.sh + d0;
    const T* knb = kb2;
    const T* vnb = vb2;
    if (fr.slot >= 0) {
        knb = reinterpret_cast<const T*>(fr.kn.ptr) + (int64_t)b * fr.kn.sb + (int64_t)hk * fr.kn.sh + d0;
        vnb = reinterpret_cast<const T*>(fr.vn.ptr) + (int64_t)b * fr.vn.sb + (int64_t)hk * fr.vn.sh + d0;
        if (split == 0 && wave == 0 && kg == 0 && dact) {      // the append: one 16-byte piece per lane
            *reinterpret_cast<u32x4*>(const_cast<T*>(kb2) + (int64_t)fr.slot * k2.sn) = *reinterpret_cast<const u32x4*>(knb);
            *reinterpret_cast<u32x4*>(const_cast<T*>(vb2) + (int64_t)fr.slot * v2.sn) = *reinterpret_cast<const u32x4*>(vnb);
        }
    }

    for (int h0 = 0; h0 < g; h0 += GT) {
        float qf[GT][EPL];
        float m[GT], l[GT], acc[GT][EPL];
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            const int h = hk * g + h0 + t;
            const T* qp = reinterpret_cast<const T*>(q.ptr) + (int64_t)b * q.sb + (int64_t)h * q.sh + d0;
            const u32x4 raw = *reinterpret_cast<const u32x4*>(qp);
            cvt16B<T, EPL>(raw, qf[t]);
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                qf[t][e] = dact ? qf[t][e] * scale : 0.f;
                acc[t][e] = 0.f;
            }
            m[t] = -INFINITY;
            l[t] = 0.f;
        }

        for (int base = k_beg; base < k_end; base += NSTREAM * U) {
            float kf[U][EPL], vf[U][EPL];
            bool valid[U];
            u32x4 kraw[U], vraw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = base + u * NSTREAM + sid;
                valid[u] = kk < k_end;
                const int kc = valid[u] ? kk : (k_end - 1);
                const bool seg1 = kc < N1;
                const bool fresh = kc - N1 == fr.slot;       // fr.slot < 0: never
                const T* kp = seg1 ? kb + (int64_t)kc * k.sn : (fresh ? knb : kb2 + (int64_t)(kc - N1) * k2.sn);
                const T* vp = seg1 ? vb + (int64_t)kc * v.sn : (fresh ? vnb : vb2 + (int64_t)(kc - N1) * v2.sn);
                kraw[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kp));
                vraw[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vp));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                cvt16B<T, EPL>(kraw[u], kf[u]);
                cvt16B<T, EPL>(vraw[u], vf[u]);
            }
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                float s[U];
                float bm = -INFINITY;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float d = 0.f;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) d = fmaf(qf[t][e], kf[u][e], d);
                    d = group_sum<LPK>(d);
                    s[u] = valid[u] ? d : -INFINITY;
                    bm = fmaxf(bm, s[u]);
                }
                const float m_new = fmaxf(m[t], bm);
                if (m_new > -INFINITY) {
                    const float alpha = __expf(m[t] - m_new);  // exp(-inf) = 0 for the first keys
                    float lt = l[t] * alpha;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) acc[t][e] *= alpha;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float pu = __expf(s[u] - m_new);
                        lt += pu;
#pragma unroll
                        for (int e = 0; e < EPL; ++e) acc[t][e] = fmaf(pu, vf[u][e], acc[t][e]);
                    }
                    l[t] = lt;
                    m[t] = m_new;
                }
            }
        }

        // merge the KPW key groups of the wave
#pragma unroll
        for (int t = 0; t < GT; ++t) {
#pragma unroll
            for (int off = LPK; off < 64; off <<= 1) {
                const float mo = __shfl_xor(m[t], off, 64);
                const float lo = __shfl_xor(l[t], off, 64);
                const float mn = fmaxf(m[t], mo);
                const float a = (m[t] == -INFINITY) ? 0.f : __expf(m[t] - mn);
                const float c = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
                l[t] = l[t] * a + lo * c;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const float ao = __shfl_xor(acc[t][e], off, 64);
                    acc[t][e] = acc[t][e] * a + ao * c;
                }
                m[t] = mn;
            }
            if (kg == 0) {
                if (chunk == 0) {
                    sm[wave][t] = m[t];
                    sl[wave][t] = l[t];
                }
#pragma unroll
                for (int e = 0; e < EPL; ++e) so[wave][t][chunk * EPL + e] = acc[t][e];
            }
        }
        __syncthreads();
        // merge the 4 waves and write the split's partial
        if (wave == 0 && kg == 0) {
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                float mn = -INFINITY;
#pragma unroll
                for (int w = 0; w < 4; ++w) mn = fmaxf(mn, sm[w][t]);
                float lt = 0.f, o[EPL];
#pragma unroll
                for (int e = 0; e < EPL; ++e) o[e] = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float a = (sm[w][t] == -INFINITY) ? 0.f : __expf(sm[w][t] - mn);
                    lt += sl[w][t] * a;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) o[e] += so[w][t][chunk * EPL + e] * a;
                }
                const int h = hk * g + h0 + t;
                const int64_t pidx = ((int64_t)b * Hq + h) * S + split;
                const bool coh = op1.cnt != nullptr;
                if (chunk == 0) {
                    st_part(Mp + pidx, mn, coh);
                    st_part(Lp + pidx, lt, coh);
                }
                if (dact) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) st_part(Op + pidx * D + chunk * EPL + e, o[e], coh);
                }
            }
        }
        __syncthreads();
    }
    if (op1.cnt) {
        __shared__ int last_flag[2];
        // this workgroup's partials went out as write-through (sc1) stores: once they are acknowledged they are visible to
        // every XCD, no L2 write-back needed (round 2 had __threadfence() on both sides of the counter: every workgroup
        // flushed and invalidated its XCD's L2, which cost more than the second launch it saved)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(&op1.cnt[b * Hkv + hk], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag[0] = old == S - 1;
            if (old == S - 1) __hip_atomic_store(&op1.cnt[b * Hkv + hk], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero for the next call
            const int total = (int)(gridDim.x * gridDim.y * gridDim.z);
            const int oldg = __hip_atomic_fetch_add(&op1.cnt[gridDim.z * Hkv], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag[1] = oldg == total - 1;
            if (oldg == total - 1) __hip_atomic_store(&op1.cnt[gridDim.z * Hkv], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (last_flag[0]) {
            // (the partials are fetched with sc1 loads: they cannot hit stale lines of this XCD's L2)
            if (D <= 128 && g > 4) {       // two heads per wave (a half-wave covers 32 x 4 columns): the group in one round up to 8 heads
                for (int hh = wave * 2 + (lane >> 5); hh < g; hh += 8)
                    reduce_head<T, true, 32>(Mp, Lp, Op, op1.s_aux, op1.o, Hq, S, D, b, hk * g + hh, lane & 31, 32);
            } else {
                for (int hh = wave; hh < g; hh += 4) reduce_head<T, true, 64>(Mp, Lp, Op, op1.s_aux, op1.o, Hq, S, D, b, hk * g + hh, lane, 64);
            }
        }
        if (last_flag[1] && fr.dyn && threadIdx.x == 0) {  // every workgroup has read the state: advance it
            const int wl = fr.dyn[1], wp = fr.dyn[2];
            fr.dyn[1] = wl + 1 < fr.wsize ? wl + 1 : fr.wsize;
            fr.dyn[2] = wp + 1 == fr.wsize ? 0 : wp + 1;
        }
    }
}

// Fold the split partials and the s_aux virtual split (decode_kernel.py:205-226).
template <typename T>
__global__ __launch_bounds__(128) void decode_reduce_kernel(const float* __restrict__ Mp, const float* __restrict__ Lp,
                                                           const float* __restrict__ Op,
                                                           const float* __restrict__ s_aux, View o, int Hq, int S,
                                                           int D, int* dyn, int wsize) {
    const int h = blockIdx.x, b = blockIdx.y;
    if (dyn && h == 0 && b == 0 && threadIdx.x == 0) {      // every reader of the state (split kernel) has finished
        const int wl = dyn[1], wp = dyn[2];
        dyn[1] = wl + 1 < wsize ? wl + 1 : wsize;
        dyn[2] = wp + 1 == wsize ? 0 : wp + 1;
    }
    reduce_head<T>(Mp, Lp, Op, s_aux, o, Hq, S, D, b, h, threadIdx.x, 128);
}

template <typename T, int LPK>
int launch_split(int gt, dim3 grid, hipStream_t stream, View q, View k, View v, View k2, View v2, int N1, Fresh fr,
                 OnePass op1, float* Mp,
                 float* Lp, float* Op, int Hq, int Hkv, int Nkv, int D, int kps, int S, float scale) {
    switch (gt) {
        case 8:
            decode_split_kernel<T, LPK, 8><<<grid, 256, 0, stream>>>(q, k, v, k2, v2, N1, fr, op1, Mp, Lp, Op, Hq, Hkv, Nkv, D, kps, S, scale);
            break;
        case 4:
            decode_split_kernel<T, LPK, 4><<<grid, 256, 0, stream>>>(q, k, v, k2, v2, N1, fr, op1, Mp, Lp, Op, Hq, Hkv, Nkv, D, kps, S, scale);
            break;
        default:
            decode_split_kernel<T, LPK, 1><<<grid, 256, 0, stream>>>(q, k, v, k2, v2, N1, fr, op1, Mp, Lp, Op, Hq, Hkv, Nkv, D, kps, S, scale);
            break;
    }
    return launch_status("decode_split");
}

template <typename T>
int launch_split_lpk(const DecodePlan& pl, dim3 grid, hipStream_t stream, View q, View k, View v, View k2, View v2,
                     int N1, Fresh fr, OnePass op1, float* Mp, float* Lp, float* Op, int Hq, int Hkv, int Nkv, int D, float scale) {
#define SFA_LPK_CASE(L)                                                                                        \
    case L:                                                                                                    \
        return launch_split<T, L>(pl.gt, grid, stream, q, k, v, k2, v2, N1, fr, op1, Mp, Lp, Op, Hq, Hkv, Nkv, D, \
                                  pl.keys_per_split, pl.splits, scale);
    switch (pl.lpk) {
        SFA_LPK_CASE(2)
        SFA_LPK_CASE(4)
        SFA_LPK_CASE(8)
        SFA_LPK_CASE(16)
        SFA_LPK_CASE(32)
        SFA_LPK_CASE(64)
    }
#undef SFA_LPK_CASE
    set_error("decode: bad lanes-per-key %d", pl.lpk);
    return SFA_ERR_UNSUPPORTED;
}

}  // namespace

int decode_plan(int64_t B, int64_t Hq, int64_t Hkv, int64_t Nkv, int64_t D, int dtype, DecodePlan* plan) {
    const int es = dtype_size(dtype);
    const int64_t row_bytes = D * es;
    if (D <= 0 || row_bytes % 16 != 0 || row_bytes > 1024) {
        set_error("decode: head dim %lld (%lld bytes/row) must be a multiple of 16 bytes and <= 1024 bytes",
                  (long long)D, (long long)row_bytes);
        return SFA_ERR_UNSUPPORTED;
    }
    if (Hkv <= 0 || Hq % Hkv != 0) {
        set_error("decode: H_q (%lld) must be divisible by H_kv (%lld)", (long long)Hq, (long long)Hkv);
        return SFA_ERR_INVALID_ARGUMENT;
    }
    const int chunks = (int)(row_bytes / 16);
    int lpk = 2;
    while (lpk < chunks) lpk <<= 1;
    const int64_t g = Hq / Hkv;
    const int gt = (g % 8 == 0) ? 8 : (g % 4 == 0) ? 4 : 1;
    const int U = gt <= 4 ? 8 : 4;
    const int iter_keys = 4 * (64 / lpk) * U;
    const int64_t target_wgs = 2048;
    const int64_t base = B * Hkv > 0 ? B * Hkv : 1;
    int64_t want = cdiv64(target_wgs, base);
    const int64_t min_keys = iter_keys > 256 ? iter_keys : 256;
    int64_t max_splits = cdiv64(Nkv, min_keys);
    if (max_splits < 1) max_splits = 1;
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    int64_t kps = cdiv64(cdiv64(Nkv > 0 ? Nkv : 1, want), iter_keys) * iter_keys;
    int64_t splits = cdiv64(Nkv > 0 ? Nkv : 1, kps);
    plan->splits = (int)splits;
    plan->max_splits = (int)want;          // splits <= want for every key count <= Nkv (want is monotonic in Nkv)
    plan->keys_per_split = (int)kps;
    plan->lpk = lpk;
    plan->gt = gt;
    return SFA_OK;
}

int decode_launch(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, int64_t n1, const sfa_tensor* k2,
                  const sfa_tensor* v2, int64_t n2, const sfa_tensor* o, const float* s_aux, void* workspace,
                  float scale, const DecodePlan& pl, hipStream_t stream, const sfa_tensor* k_new,
                  const sfa_tensor* v_new, int new_slot, int* dyn_state, bool one_pass) {
    const int B = (int)q->shape[0], Hq = (int)q->shape[1], D = (int)q->shape[3];
    const int Hkv = (int)k->shape[1], Nkv = (int)(n1 + n2), N1 = (int)n1;
    const View kv2 = k2 ? make_view(k2) : make_view(k);
    const View vv2 = v2 ? make_view(v2) : make_view(v);
    const int wsize = k2 ? (int)k2->shape[2] : 0;
    Fresh fr{kv2, vv2, -1, nullptr, wsize};
    if (k_new && v_new && (new_slot >= 0 || dyn_state))
        fr = Fresh{make_view(k_new), make_view(v_new), dyn_state ? 0 : new_slot, dyn_state, wsize};
    const int S = pl.splits;
    // counters of the one-pass mode sit at the START of the workspace (their place must not move when the split count
    // changes from one step to the next while a ring fills), the partials behind them
    float* Mp = reinterpret_cast<float*>((char*)workspace + decode_counter_bytes(B, Hkv));
    float* Lp = Mp + (int64_t)B * Hq * S;
    float* Op = Lp + (int64_t)B * Hq * S;
    one_pass = one_pass && D <= 256;      // the in-kernel fold gives a head to one wave: 64 lanes x 4 columns
    OnePass op1{one_pass ? reinterpret_cast<int*>(workspace) : nullptr, s_aux, make_view(o)};
    dim3 grid(S, Hkv, B);
    int st;
    if (q->dtype == SFA_DTYPE_F32)
        st = launch_split_lpk<float>(pl, grid, stream, make_view(q), make_view(k), make_view(v), kv2, vv2, N1, fr, op1, Mp, Lp,
                                     Op, Hq, Hkv, Nkv, D, scale);
    else if (q->dtype == SFA_DTYPE_F16)
        st = launch_split_lpk<f16_t>(pl, grid, stream, make_view(q), make_view(k), make_view(v), kv2, vv2, N1, fr, op1, Mp, Lp,
                                     Op, Hq, Hkv, Nkv, D, scale);
    else
        st = launch_split_lpk<bf16_t>(pl, grid, stream, make_view(q), make_view(k), make_view(v), kv2, vv2, N1, fr, op1, Mp, Lp,
                                      Op, Hq, Hkv, Nkv, D, scale);
    if (st != SFA_OK) return st;
    if (one_pass) {
        set_path("decode_splitkv%s_1pass_lpk%d_gt%d_s%d", dyn_state ? "_ringstep_dyn" : (fr.slot >= 0 ? "_ringstep" : (k2 ? "_ring" : "")), pl.lpk, pl.gt, S);
        return SFA_OK;
    }
    dim3 rgrid(Hq, B);
    if (q->dtype == SFA_DTYPE_F32)
        decode_reduce_kernel<float><<<rgrid, 128, 0, stream>>>(Mp, Lp, Op, s_aux, make_view(o), Hq, S, D, dyn_state, wsize);
    else if (q->dtype == SFA_DTYPE_F16)
        decode_reduce_kernel<f16_t><<<rgrid, 128, 0, stream>>>(Mp, Lp, Op, s_aux, make_view(o), Hq, S, D, dyn_state, wsize);
    else
        decode_reduce_kernel<bf16_t><<<rgrid, 128, 0, stream>>>(Mp, Lp, Op, s_aux, make_view(o), Hq, S, D, dyn_state, wsize);
    set_path("decode_splitkv%s_lpk%d_gt%d_s%d", dyn_state ? "_ringstep_dyn" : (fr.slot >= 0 ? "_ringstep" : (k2 ? "_ring" : "")), pl.lpk, pl.gt, S);
    return launch_status("decode_reduce");
}

}  // namespace sfa
