// MFMA backward for gfx950: sink + sliding-window flash attention.
// Replaces _sink_flash_attn_bwd_dkdv_kernel (sink_attention/sink_flash_attention.py:256-364) and
// _sink_flash_attn_bwd_dq_kernel (:371-484) of the reference plus its PyTorch GQA group sum (:648-651).
// Two deterministic kernels (no atomics), both recompute P = exp(S*scale - LSE) from the forward's LSE:
//
//  dK/dV kernel (K-stationary).  A workgroup = 4 wavefronts = 256 keys of one (batch, KV head); each wave owns
//    64 keys and keeps dK^T and dV^T of them in 256 accumulator registers (one wave per SIMD, 512-register waves)
//    while the workgroup sweeps ALL q heads of the GQA group x the 32-row query slices that can see the block,
//    so dK/dV are written once in [B,H_kv,N,D] (the reference writes per-Q-head copies and sums in PyTorch).
//    S and dP are computed with the KEY on the MFMA lane (S = Q K^T, dP = dO V^T with Q/dO row fragments as the
//    A operand); their accumulators, packed to 16 bit, are directly the B operands of dV^T += dO^T P and
//    dK^T += Q^T dS (A operands from ds_read_b64_tr_b16 on the same LDS image of the Q / dO slice).  -LSE/scale
//    and -Delta are loaded as the INITIAL accumulators of S and dP, so p = exp2(c * S') and dS = p * dP' need no
//    row broadcast.  Sink keys (block 0) simply sweep every later query slice; block 0 is dispatched first.
//
//  dQ kernel (Q-stationary).  Same work decomposition as the forward (8 waves = HPW q heads x row blocks, K/V
//    tile staged once per workgroup): S^T = K Q^T, dP^T = V dO^T with the query row on the lane, dS^T packed as
//    the B operand of dQ^T += K^T dS^T (K^T fragments by transposed LDS reads of the K tile).
#include "sfa_common.hpp"
#include "sfa_internal.hpp"

#include <cstdlib>
#include <type_traits>

namespace sfa {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// Diagnostic build only (-DSFA_STAMPS): s_memtime stamps around the phases of the plain dK/dV loop, summed per wave
// and dumped for one mid-grid workgroup.  Never compiled into the shipped library.
#ifdef SFA_STAMPS
__device__ unsigned long long g_stamps[8 * 8];
#define STAMP_DECL unsigned long long ts_prev_ = 0, ts_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(k)                                                                              \
    {                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        ts_acc_[k] += t_ - ts_prev_;                                                          \
        ts_prev_ = t_;                                                                        \
    }
#else
#define STAMP_DECL
#define STAMP(k)
#endif

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    using frag = bf16x8_t;
    using elem = __bf16;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<f16_t> {
    using frag = f16x8_t;
    using elem = _Float16;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

struct BwdArgs {
    View q, k, v, d_o, dq, dk, dv;
    const float* lse;
    const float* delta;
    const float* consts;   // [B, Hq, 2, N] f32: -LSE*log2e | -Delta (row constants of the wave-specialised dK/dV kernel)
    int B, Hq, Hkv, N;
    int num_sink, window;  // window clamped to [0, N]
    float scale, scale_log2;
    unsigned q_range, k_range, v_range, do_range, dq_range, dk_range, dv_range;
    int n_kblocks;               // dK/dV kernel: ceil(N / 128)
    int hpw, rb, n_qtiles, hgroups;  // dQ kernel
    int prio;                        // experiment knob: score-wave priority in the wave-specialised dK/dV kernel
    // dS spill (SFA_FLAG_BWD_SPILL_DS): the dK/dV kernel saves dS (16 bit) per (head, key block, 32-row slice) and dQ
    // becomes a plain GEMM over it.  Chunk (kb, qt) of a head lives at ds + (head_idx * ds_chunks + ds_prefix[kb] +
    // qt - 4*kb) * kDsChunk
    char* ds;
    const int* ds_prefix;            // [n_kblocks + 1] exclusive prefix of slices per key block
    const int* cu;                   // packed batches: device row offsets [B + 1] (else null), see Problem
    int n_total;                     // rows of the packed tensors
    int Nk;                          // key rows (>= N): query row i sits at position i + Nk - N
    int consts_ready;                // the preprocess pass already wrote the row constants
    int ds_nt;                       // tuning knob (SFA_DS_NT): non-temporal hint on the dS stream loads of the dQ GEMM
    int64_t ds_chunks;               // chunks per head = ds_prefix[n_kblocks]
};

constexpr int kDsChunk = 32 * 128 * 2;   // bytes of one dS chunk: 32 query rows x 128 keys x 16 bit

// query slices (32 rows) that the 128-key block kb must sweep: [4*kb, 4*kb + nq)
__host__ __device__ inline int dkdv_slices(int kb, int N, int ns, int W) {
    const int kb0 = kb * 128;
    const int kb1 = (kb0 + 128 < N) ? kb0 + 128 : N;
    int i_hi;
    if (kb0 < ns) {
        i_hi = N;
    } else {
        i_hi = kb1 - 1 + W;
        if (i_hi > N) i_hi = N;
    }
    const int qt_lo = kb0 / 32;
    int qt_hi = (i_hi + 31) / 32;
    if (qt_hi < qt_lo) qt_hi = qt_lo;
    return qt_hi - qt_lo;
}

// One LDS image serves row reads (ds_read_b128 MFMA operand) and transposed reads (ds_read_b64_tr_b16):
// XOR the 16-byte chunk index with sw(row).  256-byte rows: 16 chunks; 128-byte rows (D <= 64): 8 chunks.
template <int ROWB>
__device__ __forceinline__ int sw(int row) {
    if constexpr (ROWB == 256)
        return ((row & 3) << 2) | ((row >> 2) & 3);
    else
        return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}

template <typename frag>
__device__ __forceinline__ frag tr_pair(const char* p, int rowb8) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + rowb8));
    s16x8 vv;
    vv[0] = lo[0]; vv[1] = lo[1]; vv[2] = lo[2]; vv[3] = lo[3];
    vv[4] = hi[0]; vv[5] = hi[1]; vv[6] = hi[2]; vv[7] = hi[3];
    return __builtin_bit_cast(frag, vv);
}

template <typename frag>
__device__ __forceinline__ frag tr_pair2(const char* p1, const char* p2) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p2));
    s16x8 vv;
    vv[0] = lo[0]; vv[1] = lo[1]; vv[2] = lo[2]; vv[3] = lo[3];
    vv[4] = hi[0]; vv[5] = hi[1]; vv[6] = hi[2]; vv[7] = hi[3];
    return __builtin_bit_cast(frag, vv);
}

// Transposed LDS read as inline asm with an immediate offset.  hipcc puts s_waitcnt vmcnt(0) in front of the
// ds_read_tr BUILTIN whenever an LDS-DMA is in flight (may-alias), which would drain the slice prefetch every trip;
// an asm read is invisible to that pass.  The caller waits lgkmcnt(0) + sched_barrier(0) before the first use.
template <int OFF>
__device__ __forceinline__ s16x4 tr_read_asm(unsigned addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
    return v;
}
__device__ __forceinline__ s16x8 join8(s16x4 lo, s16x4 hi) {
    s16x8 vv;
    vv[0] = lo[0]; vv[1] = lo[1]; vv[2] = lo[2]; vv[3] = lo[3];
    vv[4] = hi[0]; vv[5] = hi[1]; vv[6] = hi[2]; vv[7] = hi[3];
    return vv;
}

// ===================================================================== dK / dV
constexpr int kKB = 128;   // keys per workgroup (32 per wave)

template <typename T, int D>
__global__ __launch_bounds__(256, 1) void bwd_dkdv_mfma_kernel(BwdArgs a) {
    using M = Mma<T>;
    using frag = typename M::frag;
    using E = typename M::elem;
    constexpr int DK = D / 16;
    constexpr int DVB = (D + 31) / 32;
    constexpr int CPR = D / 8;
    constexpr int ROWB = (D <= 64) ? 128 : 256;
    constexpr int QT = 64;                       // query rows per iteration (two 32-row MFMA slices)
    constexpr int TILE = QT * ROWB;              // bytes of one Q (or dO) slice image
    constexpr int STAGE = 2 * TILE + 512;        // Q | dO | -lse*log2e[64] | -delta[64]
    constexpr int NCH = QT * CPR;                // 16-byte chunks per slice
    constexpr int NLD = (NCH + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int kb = bid % a.n_kblocks;
    int rest = bid / a.n_kblocks;
    const int hk = rest % a.Hkv;
    const int b = rest / a.Hkv;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int N = a.N, W = a.window, ns = a.num_sink;
    const int g = a.Hq / a.Hkv;
    const int kb0 = kb * kKB;
    const int kb1 = (kb0 + kKB < N) ? kb0 + kKB : N;
    const int kw0 = kb0 + 32 * wave;   // wave's first key
    const int key = kw0 + r;           // lane's key

    const char* kbase = a.k.ptr + ((int64_t)b * a.k.sb + (int64_t)hk * a.k.sh) * 2;
    const char* vbase = a.v.ptr + ((int64_t)b * a.v.sb + (int64_t)hk * a.v.sh) * 2;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, a.k_range, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, a.v_range, 0x00020000);

    // ---- K and V fragments of the wave's 32 keys stay in registers: B operands of S = Q K^T, dP = dO V^T
    frag kf[DK], vf[DK];
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) {
        const unsigned ko = (unsigned)key * (unsigned)(a.k.sn * 2) + (unsigned)((2 * ks + h) * 16);
        const unsigned vo = (unsigned)key * (unsigned)(a.v.sn * 2) + (unsigned)((2 * ks + h) * 16);
        kf[ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rk, ko, 0, 0));
        vf[ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rv, vo, 0, 0));
    }
    // loop-invariant MFMA B operands: pin them in the accumulator half of the register file (frees 64 VGPRs)
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) {
        u32x4 kx = __builtin_bit_cast(u32x4, kf[ks]), vx = __builtin_bit_cast(u32x4, vf[ks]);
        asm volatile("" : "+a"(kx), "+a"(vx));
        kf[ks] = __builtin_bit_cast(frag, kx);
        vf[ks] = __builtin_bit_cast(frag, vx);
    }

    // ---- iteration space: q heads of the group x 64-row slices that can see the block
    int i_hi;
    if (kb0 < ns) {
        i_hi = N;
    } else {
        i_hi = kb1 - 1 + W;
        if (i_hi > N) i_hi = N;
    }
    const int qt_lo = kb0 / QT;
    int qt_hi = (i_hi + QT - 1) / QT;
    if (qt_hi < qt_lo) qt_hi = qt_lo;
    const int nq = qt_hi - qt_lo;
    const int n_it = g * nq;

    f32x16 dKt[DVB], dVt[DVB];
#pragma unroll
    for (int db = 0; db < DVB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            dKt[db][i] = 0.f;
            dVt[db][i] = 0.f;
        }

    u32x4 qst[NLD], dst[NLD];
    float cst = 0.f;
    bool cst_oob = false;
    auto issue_loads = [&](int it) {
        const int hh = it / nq, qt = qt_lo + it % nq;
        const int head = hk * g + hh;
        const char* qb = a.q.ptr + ((int64_t)b * a.q.sb + (int64_t)head * a.q.sh) * 2;
        const char* dob = a.d_o.ptr + ((int64_t)b * a.d_o.sb + (int64_t)head * a.d_o.sh) * 2;
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)qb, 0, a.q_range, 0x00020000);
        const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc((void*)dob, 0, a.do_range, 0x00020000);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * 256;
            const int row = c / CPR, ch = c % CPR;
            const bool in = (NCH % 256 == 0) || (c < NCH);
            const unsigned qo = in ? (unsigned)(qt * QT + row) * (unsigned)(a.q.sn * 2) + (unsigned)(ch * 16) : 0xFFFFFFF0u;
            const unsigned oo = in ? (unsigned)(qt * QT + row) * (unsigned)(a.d_o.sn * 2) + (unsigned)(ch * 16) : 0xFFFFFFF0u;
            qst[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, qo, 0, 0);
            dst[i] = __builtin_amdgcn_raw_buffer_load_b128(rdo, oo, 0, 0);
        }
        if (tid < 128) {   // raw value only: converting here would wait a full memory latency inside the loop
            const int row = qt * QT + (tid & 63);
            const int64_t idx = ((int64_t)b * a.Hq + head) * N + (row < N ? row : N - 1);
            cst = (tid < 64) ? a.lse[idx] : a.delta[idx];
            cst_oob = row >= N;
        }
    };
    auto write_lds = [&](int buf) {
        char* st = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * 256;
            const int row = c / CPR, ch = c % CPR;
            if ((NCH % 256 == 0) || (c < NCH)) {
                const int o = row * ROWB + ((ch ^ sw<ROWB>(row)) << 4);
                *reinterpret_cast<u32x4*>(st + o) = qst[i];
                *reinterpret_cast<u32x4*>(st + TILE + o) = dst[i];
            }
        }
        if (tid < 128)   // -LSE*log2e (rows >= N: -inf => p = 0) | -Delta
            *reinterpret_cast<float*>(st + 2 * TILE + tid * 4) =
                (tid < 64) ? (cst_oob ? -INFINITY : -cst * kLog2e) : (cst_oob ? 0.f : -cst);
    };

    // per-lane LDS addressing
    const int rowrd = r * ROWB;                 // row read of slice row (32*sub + r): A operand of S / dP
    const int rsw = sw<ROWB>(r);                // sw(32*sub + r) == sw(r)
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const int tr_row = 4 * h + q4;              // + 32*sub + 16*s ; second read + 8
    const int tr_col = 2 * g1 + (p4 >> 1);      // + 4*db
    const int tr_byte = (p4 & 1) * 8;
    const float c = a.scale_log2;

    if (n_it > 0) {
        issue_loads(0);
        write_lds(0);
    }
    __syncthreads();

    STAMP_DECL
    for (int it = 0; it < n_it; ++it) {
        const int buf = it & 1;
        STAMP(7)
        if (it + 1 < n_it) issue_loads(it + 1);
        STAMP(0)
        const int q0 = (qt_lo + it % nq) * QT;
        // wave-level classification of (32 keys) x (64 rows).  The MFMAs run unconditionally (a tile the wave
        // does not need is simply an all-masked tile): no control flow around the accumulators.
        const bool full = (kw0 + 31 <= q0) && (q0 + 63 < N) && ((kw0 + 31 < ns) || (kw0 + W > q0 + 63));
        {
            const char* st = smem + buf * STAGE;
            const char* ql = st;
            const char* dol = st + TILE;
            const float* cl = reinterpret_cast<const float*>(st + 2 * TILE);
            frag pP[2][2], pS[2][2];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16 S, dP;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    S[i] = 0.f;
                    dP[i] = 0.f;
                }
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) {
                    const int o = sub * 32 * ROWB + rowrd + (((2 * ks + h) ^ rsw) << 4);
                    const frag qa = *reinterpret_cast<const frag*>(ql + o);
                    const frag da = *reinterpret_cast<const frag*>(dol + o);
                    S = M::run(qa, kf[ks], S);
                    dP = M::run(da, vf[ks], dP);
                }
                STAMP(1)
                // row constants (row = query (i&3)+8(i>>2)+4h): nl = -LSE*log2e, nd = -Delta
                float nl[16], nd[16];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(cl + 32 * sub + 8 * g4 + 4 * h);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(cl + 64 + 32 * sub + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        nl[4 * g4 + e] = l4[e];
                        nd[4 * g4 + e] = d4[e];
                    }
                }
                // P = exp2(c*S - LSE*log2e), dS = P * (dP - Delta) -> packed B operands (k index = query row, permuted)
                if (full) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(S[i], c, nl[i]));
                        pP[sub][i >> 3][i & 7] = (E)p;
                        pS[sub][i >> 3][i & 7] = (E)(p * (dP[i] + nd[i]));
                    }
                } else {
                    int keym = key;
                    asm volatile("; edge tile" : "+v"(keym) : : "memory");   // side effect: the branch cannot be speculated / if-converted
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int qi = q0 + 32 * sub + (i & 3) + 8 * (i >> 2) + 4 * h;
                        const bool valid = (keym <= qi) && (keym < ns || keym + W > qi) && (qi < N);
                        float p = __builtin_amdgcn_exp2f(fmaf(S[i], c, nl[i]));
                        p = valid ? p : 0.f;                      // select, not multiply: p may be inf on masked slots
                        pP[sub][i >> 3][i & 7] = (E)p;
                        pS[sub][i >> 3][i & 7] = (E)(p * (dP[i] + nd[i]));
                    }
                }
                STAMP(2)
            }
            // dV^T += dO^T P ; dK^T += Q^T dS   (A operands: transposed reads of the dO / Q slice images)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int db = 0; db < DVB; ++db) {
                        const int row = 32 * sub + 16 * s + tr_row;
                        const int o1 = row * ROWB + (((4 * db + tr_col) ^ sw<ROWB>(row)) << 4) + tr_byte;
                        const int o2 = (row + 8) * ROWB + (((4 * db + tr_col) ^ sw<ROWB>(row + 8)) << 4) + tr_byte;
                        dVt[db] = M::run(tr_pair2<frag>(dol + o1, dol + o2), pP[sub][s], dVt[db]);
                        dKt[db] = M::run(tr_pair2<frag>(ql + o1, ql + o2), pS[sub][s], dKt[db]);
                    }
        }
        STAMP(3)
        if (it + 1 < n_it) write_lds(buf ^ 1);
        STAMP(4)
        __syncthreads();
        STAMP(5)
    }
#ifdef SFA_STAMPS
    if (kb == 40 && hk == 0 && b == 0 && lane == 0 && a.num_sink < 0) {
        for (int k = 0; k < 8; ++k) g_stamps[wave * 8 + k] = ts_acc_[k];
        g_stamps[wave * 8 + 6] = (unsigned long long)n_it;
    }
#endif

    // ---- epilogue: dK = scale * dK^T^T, dV = dV^T^T ; lane = key, registers = d
    char* dkb = a.dk.ptr + ((int64_t)b * a.dk.sb + (int64_t)hk * a.dk.sh) * 2;
    char* dvb = a.dv.ptr + ((int64_t)b * a.dv.sb + (int64_t)hk * a.dv.sh) * 2;
    const __amdgpu_buffer_rsrc_t rdk = __builtin_amdgcn_make_buffer_rsrc((void*)dkb, 0, a.dk_range, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdv = __builtin_amdgcn_make_buffer_rsrc((void*)dvb, 0, a.dv_range, 0x00020000);
    typedef __attribute__((ext_vector_type(4))) E e4;
#pragma unroll
    for (int db = 0; db < DVB; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = 32 * db + 8 * g4 + 4 * h;
            if (d < D) {
                e4 pk, pv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pk[e] = (E)(dKt[db][4 * g4 + e] * a.scale);
                    pv[e] = (E)(dVt[db][4 * g4 + e]);
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rdk,
                                                      (unsigned)key * (unsigned)(a.dk.sn * 2) + (unsigned)(d * 2), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pv), rdv,
                                                      (unsigned)key * (unsigned)(a.dv.sn * 2) + (unsigned)(d * 2), 0, 0);
            }
        }
}

// ================================================ dK / dV, wave-specialised (8 waves)
// Two waves per SIMD without blowing the register budget: the work of a 32-key group is split by ROLE.
//   score wave  (waves 0-3): S = Q K^T, dP = dO V^T (16 MFMAs per 32-row slice), P = exp2(..), dS = P*(dP-Delta),
//                packs P and dS into MFMA B fragments and drops them in an LDS exchange buffer.  Its K/V fragments
//                are pinned in AGPRs; no big accumulators.
//   accumulate wave (waves 4-7, same SIMD as wave-4): owns dK^T/dV^T of the group in 128 accumulators; per slice
//                reads the 4 fragments + transposed Q/dO reads, 16 MFMAs.
// The accumulate wave works one slice behind the score wave: one s_barrier per 32-row trip, FOUR slice stages
// (row reads of slice t, transposed reads of slice t-1, slice t+1 landed or landing, LDS-DMA of slice t+2 being
// issued) and a double-buffered exchange.  A trip is ~1 us, about one memory latency, so the DMA runs three slices
// ahead and the trip ends on a COUNTED s_waitcnt vmcnt(2) + raw s_barrier (never vmcnt(0) in the loop).  Both roles
// issue the big LDS read burst of their NEXT trip (row fragments of slice t+1 / transposed fragments of slice t) at
// the end of the current one, so MFMAs start right after the barrier and the burst overlaps the other role's work.
// Loop hygiene (measured with s_memtime stamps, tools/stamps.py): no integer division in the loop (slice
// coordinates advance incrementally), descriptors rebuilt only when the q head changes, trips unrolled x4 so
// the stage offset of every ds_read is an immediate, per-lane swizzled offsets hoisted.
template <typename T, int D, bool SPILL>
__global__ __launch_bounds__(512, 2) void bwd_dkdv_mfma_kernel_ws(BwdArgs a) {
    using M = Mma<T>;
    using frag = typename M::frag;
    using E = typename M::elem;
    constexpr int DK = D / 16;
    constexpr int DVB = (D + 31) / 32;
    constexpr int CPR = D / 8;
    constexpr int ROWB = (D <= 64) ? 128 : 256;
    constexpr int QT = 32;
    constexpr int TILE = QT * ROWB;
    constexpr int STAGE = 2 * TILE + 256;            // Q | dO | -lse*log2e[32] | -delta[32]
    constexpr int CPRD = ROWB / 16;
    constexpr int NSLOT = QT * CPRD;                  // 16-byte slots per slice image (512 or 256)
    constexpr int XBYTES = 4 * 1024;                  // exchange per key group: pP[0], pP[1], pS[0], pS[1]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NST = 4;
    char* xch = smem + NST * STAGE;                   // [2 buffers][4 key groups][XBYTES]

    // Longest first inside every XCD's share: a block that holds sink keys (block 0 when num_sink > 0) sweeps EVERY
    // later query slice, twice the trips of a full window block; dispatched in plain (group, block) order the last
    // group's block 0 starts three quarters into the kernel and ends it alone.  So each XCD (a contiguous range of
    // (b, KV head) groups after xcd_remap) runs the block 0 of all its groups first, then the other blocks.
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    int kb, grp;
    {
        const int nK = a.n_kblocks, nG = (int)gridDim.x / nK;
        const int chunk = nG >> 3;                        // groups per XCD
        if (a.num_sink > 0 && nK > 1 && (nG & 7) == 0 && chunk > 0) {
            const int per = chunk * nK, xcd = bid / per, j = bid - xcd * per;
            if (j < chunk) {
                grp = xcd * chunk + j;
                kb = 0;
            } else {
                // (block-major over the XCD's groups, i.e. strictly longest-first, measured slightly slower and left
                // the caches colder for the next kernel)
                const int r2 = j - chunk;
                grp = xcd * chunk + r2 / (nK - 1);
                kb = 1 + r2 % (nK - 1);
            }
        } else {
            kb = bid % nK;
            grp = bid / nK;
        }
    }
    const int hk = grp % a.Hkv;
    const int b = grp / a.Hkv;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave & 3;
    const bool acc_role = wave >= 4;
    const int r = lane & 31, h = lane >> 5;
    const SeqInfo sq = seq_of(a.cu, b, a.N);
    const int N = sq.N, ns = a.num_sink;       // N = query rows
    const int P = a.cu ? 0 : a.Nk - a.N;       // position of query row 0 among the keys
    const int Nk = N + P;                      // key rows
    const int W = a.window < Nk ? a.window : Nk;
    const int NT_ = a.cu ? a.n_total : N;      // rows of one head in the [B, Hq, 2, rows] row-constant buffer
    const int g = a.Hq / a.Hkv;
    const int kb0 = kb * kKB;
    if (kb0 >= Nk) return;                     // packed batches: the grid is sized for the longest sequence
    const int kb1 = (kb0 + kKB < Nk) ? kb0 + kKB : Nk;
    const int kw0 = kb0 + 32 * kg;
    const int key = kw0 + r;

    // rows that can see the block: key j is visible to row i iff j <= i + P and (j < ns or j >= i + P - W + 1)
    int i_hi;
    if (kb0 < ns) {
        i_hi = N;
    } else {
        i_hi = kb1 - 1 + W - P;
        if (i_hi > N) i_hi = N;
        if (i_hi < 0) i_hi = 0;
    }
    const int qt_lo = (kb0 > P ? kb0 - P : 0) / QT;
    int qt_hi = (i_hi + QT - 1) / QT;
    if (qt_hi < qt_lo) qt_hi = qt_lo;
    const int nq = qt_hi - qt_lo;          // == dkdv_slices(kb, N, ns, W)
    const int n_it = g * nq;

    // ---- slice loader: ONLY the accumulate waves stage slices (they idle at the barrier; on the score waves the
    // ~350 cycles of counter / descriptor / M0 work per trip were critical-path time).  Next slice = (ld_hh, ld_qt);
    // per-lane source offsets are fixed, the swizzle is applied to the source chunk (the DMA fills LDS linearly).
    constexpr int NDW = NSLOT / 256;                              // DMA instructions per tensor per loader wave
    const int lw = wave - 4;                                      // loader wave index 0..3 (accumulate role)
    unsigned voq0[NDW], vod0[NDW];
#pragma unroll
    for (int i = 0; i < NDW; ++i) {
        const int cidx = i * 256 + lw * 64 + lane;
        const int drow = cidx / CPRD;
        const int dch = (cidx % CPRD) ^ sw<ROWB>(drow);
        const bool din = dch < CPR;
        voq0[i] = din ? (unsigned)drow * (unsigned)(a.q.sn * 2) + (unsigned)(dch * 16) : 0xFFFFFFF0u;
        vod0[i] = din ? (unsigned)drow * (unsigned)(a.d_o.sn * 2) + (unsigned)(dch * 16) : 0xFFFFFFF0u;
    }
    const unsigned qstep = (unsigned)(QT * a.q.sn * 2), dstep = (unsigned)(QT * a.d_o.sn * 2);
    int ld_hh = 0, ld_qt = qt_lo;
    __amdgpu_buffer_rsrc_t rq, rdo, rc;
    const unsigned voc0 = (unsigned)((lane < 32 ? 0 : NT_) + (lane & 31)) * 4u;   // wave 4: -LSE*log2e | -Delta rows
    const unsigned q_rng = seq_range(a.cu, a.q_range, N, a.q.sn, D), do_rng = seq_range(a.cu, a.do_range, N, a.d_o.sn, D);
    auto set_head = [&](int hh) {
        const int head = hk * g + hh;
        const char* qb = a.q.ptr + ((int64_t)sq.bb * a.q.sb + (int64_t)head * a.q.sh + (int64_t)sq.row0 * a.q.sn) * 2;
        const char* dob = a.d_o.ptr + ((int64_t)sq.bb * a.d_o.sb + (int64_t)head * a.d_o.sh + (int64_t)sq.row0 * a.d_o.sn) * 2;
        rq = __builtin_amdgcn_make_buffer_rsrc((void*)qb, 0, q_rng, 0x00020000);
        rdo = __builtin_amdgcn_make_buffer_rsrc((void*)dob, 0, do_rng, 0x00020000);
        rc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.consts + ((int64_t)sq.bb * a.Hq + head) * 2 * NT_ + sq.row0), 0,
                                               (unsigned)((2 * NT_ - sq.row0) * 4), 0x00020000);
    };
    auto stage_next = [&](int stage_off) {     // stage_off = byte offset of the destination stage (compile-time at call sites)
        if (!acc_role) return;
        if (ld_qt == qt_lo) set_head(ld_hh);
        // the loop issues nothing but LDS-DMA: an ordinary global load next to in-flight DMAs makes hipcc drain
        // vmcnt(0) first.  Wave 4 also brings the slice's 64 row constants (rows >= N of the Delta half read 0; of
        // the LSE half they read finite garbage that only edge tiles can see, and those mask by row < N).
        if (lw == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (__attribute__((address_space(3))) void*)(smem + stage_off + 2 * TILE),
                                                     4, voc0 + (unsigned)ld_qt * (unsigned)(QT * 4), 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NDW; ++i) {
            const unsigned qo = voq0[i] == 0xFFFFFFF0u ? voq0[i] : voq0[i] + (unsigned)ld_qt * qstep;
            const unsigned oo = vod0[i] == 0xFFFFFFF0u ? vod0[i] : vod0[i] + (unsigned)ld_qt * dstep;
            char* dstq = smem + stage_off + (i * 256 + lw * 64) * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (__attribute__((address_space(3))) void*)dstq, 16, qo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdo, (__attribute__((address_space(3))) void*)(dstq + TILE), 16, oo, 0, 0, 0);
        }
        if (++ld_qt == qt_hi) {
            ld_qt = qt_lo;
            ++ld_hh;
        }
    };

    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const float c = a.scale_log2;

    // end of a trip: on the loader waves everything but the DMAs just issued (2*NDW, +1 on wave 4) must have
    // landed; exchange writes / fragment reads done
    // (dS spill: the accumulate wave's two dS stores of this trip, issued after the DMAs, may stay in flight too)
    auto trip_sync = [&](bool more, bool stored = false) {
        if (!acc_role || !more)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (lw == 0 && stored)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NDW + 3) : "memory");
        else if (lw == 0)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NDW + 1) : "memory");
        else if (stored)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NDW + 2) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NDW) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);   // nothing (MFMAs on prefetched fragments included) moves above the barrier
    };
    if (n_it > 0) {
        stage_next(0);
        if (n_it > 1) stage_next(STAGE);
        if (n_it > 2) stage_next(2 * STAGE);
    }
    __syncthreads();
    const int n_trips = n_it + 1;

    if (!acc_role) {
        // ------------------------------------------------------------------ score waves
        // the score wave is the critical path of a trip: let it win MFMA/VALU arbitration on its SIMD so the
        // accumulate wave's MFMAs fill its exp/pack phase instead of delaying its S/dP chain
        if (a.prio) __builtin_amdgcn_s_setprio(2);
        const char* kbase = a.k.ptr + ((int64_t)sq.bb * a.k.sb + (int64_t)hk * a.k.sh + (int64_t)sq.row0 * a.k.sn) * 2;
        const char* vbase = a.v.ptr + ((int64_t)sq.bb * a.v.sb + (int64_t)hk * a.v.sh + (int64_t)sq.row0 * a.v.sn) * 2;
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, seq_range(a.cu, a.k_range, Nk, a.k.sn, D), 0x00020000);
        const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, seq_range(a.cu, a.v_range, Nk, a.v.sn, D), 0x00020000);
        frag kf[DK], vf[DK];
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
            const unsigned ko = (unsigned)key * (unsigned)(a.k.sn * 2) + (unsigned)((2 * ks + h) * 16);
            const unsigned vo = (unsigned)key * (unsigned)(a.v.sn * 2) + (unsigned)((2 * ks + h) * 16);
            kf[ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rk, ko, 0, 0));
            vf[ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rv, vo, 0, 0));
        }
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) {
            u32x4 kx = __builtin_bit_cast(u32x4, kf[ks]), vx = __builtin_bit_cast(u32x4, vf[ks]);
            asm volatile("" : "+a"(kx), "+a"(vx));
            kf[ks] = __builtin_bit_cast(frag, kx);
            vf[ks] = __builtin_bit_cast(frag, vx);
        }
        // hoisted per-lane LDS offsets of the row reads (one VGPR per k-step; stage / tile go into the immediate)
        int rdo_[DK];
#pragma unroll
        for (int ks = 0; ks < DK; ++ks) rdo_[ks] = r * ROWB + (((2 * ks + h) ^ sw<ROWB>(r)) << 4);
        const char* xw = xch + kg * XBYTES + lane * 16;
        int q0 = qt_lo * QT;          // first row of the slice being scored
        int sc_q = 0;                 // slices scored for the current head

        STAMP_DECL
        // operand fragments + row constants of the slice to score next (filled one trip ahead)
        frag qa[DK], da[DK];
        f32x4 l4[4], d4[4];
        auto prefetch_q = [&](auto stage_tag) {
            const char* ql = smem + decltype(stage_tag)::value * STAGE;
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) qa[ks] = *reinterpret_cast<const frag*>(ql + rdo_[ks]);
        };
        auto prefetch_d = [&](auto stage_tag) {
            const char* dol = smem + decltype(stage_tag)::value * STAGE + TILE;
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) da[ks] = *reinterpret_cast<const frag*>(dol + rdo_[ks]);
        };
        auto prefetch_c = [&](auto stage_tag) {
            const float* cl = reinterpret_cast<const float*>(smem + decltype(stage_tag)::value * STAGE + 2 * TILE);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                l4[g4] = *reinterpret_cast<const f32x4*>(cl + 8 * g4 + 4 * h);
                d4[g4] = *reinterpret_cast<const f32x4*>(cl + 32 + 8 * g4 + 4 * h);
            }
        };
        // Q fragments of slice t+1 (stage next_tag, resident since the end of trip t-1; past the last slice the stage
        // is allocated but stale and the fragments unused) are fetched right after the S chain has consumed the
        // current ones, so the S chain of the next trip starts straight after the barrier; the dO fragments and row
        // constants of the current slice are fetched at trip start, under the S chain (no registers to hold more)
        auto score = [&](auto full_tag, auto cur_tag, auto next_tag, int par) {
            constexpr bool FULL = decltype(full_tag)::value;
            prefetch_d(cur_tag);
            prefetch_c(cur_tag);
            STAMP(1)
            f32x16 S, dP;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                S[i] = 0.f;
                dP[i] = 0.f;
            }
            // S chain first; its exp2 work is then issued in the shadow of the dP chain (independent of it)
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) S = M::run(qa[ks], kf[ks], S);
            prefetch_q(next_tag);
            float pf[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * g4 + e;
                    float p = __builtin_amdgcn_exp2f(fmaf(S[i], c, l4[g4][e]));
                    if constexpr (!FULL) {
                        const int qi = q0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        const bool valid = (key <= qi + P) && (key < ns || key + W > qi + P) && (qi < N);
                        p = valid ? p : 0.f;
                    }
                    pf[i] = p;
                }
            STAMP(2)
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) dP = M::run(da[ks], vf[ks], dP);
            if constexpr (FULL) {
                // 1 MFMA then a few VALU/transcendental ops, repeated: the p = exp2(..) work fills the dP chain
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // VALU
                }
            }
            frag pP[2], pS[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                pP[i >> 3][i & 7] = (E)pf[i];
                pS[i >> 3][i & 7] = (E)(pf[i] * (dP[i] + d4[i >> 2][i & 3]));
            }
            STAMP(3)
            char* x = const_cast<char*>(xw) + par * (4 * XBYTES);
            *reinterpret_cast<frag*>(x) = pP[0];
            *reinterpret_cast<frag*>(x + 1024) = pP[1];
            *reinterpret_cast<frag*>(x + 2048) = pS[0];
            *reinterpret_cast<frag*>(x + 3072) = pS[1];
        };
        // trip t (t % 4 == K): DMA slice t+2 into stage (K+2)%4, score slice t from stage K
        auto trip = [&](auto ktag, int t) {
            constexpr int K = decltype(ktag)::value;
            STAMP(7)
            STAMP(0)
            if (t < n_it) {
                const bool full = (kw0 + 31 <= q0 + P) && (q0 + 31 < N) && ((kw0 + 31 < ns) || (kw0 + W > q0 + 31 + P));
                if (full)
                    score(std::true_type{}, ktag, std::integral_constant<int, (K + 1) % NST>{}, t & 1);
                else
                    score(std::false_type{}, ktag, std::integral_constant<int, (K + 1) % NST>{}, t & 1);
                q0 += QT;
                if (++sc_q == nq) {
                    sc_q = 0;
                    q0 = qt_lo * QT;
                }
            }
            STAMP(4)
            trip_sync(t + 3 < n_it);
            STAMP(5)
        };
        if (n_it > 0) prefetch_q(std::integral_constant<int, 0>{});
        int t = 0;
        for (; t + 4 <= n_trips; t += 4) {
            trip(std::integral_constant<int, 0>{}, t);
            trip(std::integral_constant<int, 1>{}, t + 1);
            trip(std::integral_constant<int, 2>{}, t + 2);
            trip(std::integral_constant<int, 3>{}, t + 3);
        }
        if (t < n_trips) trip(std::integral_constant<int, 0>{}, t++);
        if (t < n_trips) trip(std::integral_constant<int, 1>{}, t++);
        if (t < n_trips) trip(std::integral_constant<int, 2>{}, t++);
#ifdef SFA_STAMPS
        if (kb == 40 && hk == 0 && b == 0 && lane == 0) {
            for (int kk = 0; kk < 8; ++kk) g_stamps[wave * 8 + kk] = ts_acc_[kk];
            g_stamps[wave * 8 + 6] = (unsigned long long)n_it;
        }
#endif
    } else {
        // ------------------------------------------------------------------ accumulate waves
        f32x16 dKt[DVB], dVt[DVB];
#pragma unroll
        for (int db = 0; db < DVB; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                dKt[db][i] = 0.f;
                dVt[db][i] = 0.f;
            }
        // hoisted per-lane offsets of the transposed reads: [db][first / +8 rows]; s (16 rows) and stage are immediates
        const int tr_row = 4 * h + q4;
        const int tr_col = 2 * g1 + (p4 >> 1);
        const int tr_byte = (p4 & 1) * 8;
        const unsigned smem_base = (unsigned)(size_t)smem;   // LDS byte address of the dynamic region
        unsigned ta1[DVB], ta2[DVB];
#pragma unroll
        for (int db = 0; db < DVB; ++db) {
            ta1[db] = smem_base + (unsigned)(tr_row * ROWB + (((4 * db + tr_col) ^ sw<ROWB>(tr_row)) << 4) + tr_byte);
            ta2[db] = smem_base + (unsigned)((tr_row + 8) * ROWB + (((4 * db + tr_col) ^ sw<ROWB>(tr_row + 8)) << 4) + tr_byte);
        }
        const char* xr = xch + kg * XBYTES + lane * 16;
        // dS spill: the accumulate wave (it has the slack) also saves the dS fragments it reads from the exchange:
        // per key group 2 KB = [pS index s2][lane][16 B], so each store instruction writes 1 KB contiguous
        char* dsp = nullptr;
        int64_t ds_head_step = 0;
        int ac_q = 0;                 // slices accumulated for the current head
        u32x4 sv0 = {0, 0, 0, 0}, sv1 = {0, 0, 0, 0};
        if constexpr (SPILL) {
            dsp = a.ds + (((int64_t)b * a.Hq + (int64_t)hk * g) * a.ds_chunks + a.ds_prefix[kb]) * kDsChunk + kg * 2048 +
                  lane * 16;
            ds_head_step = (a.ds_chunks - nq) * (int64_t)kDsChunk;
        }
        // transposed fragments of the slice to accumulate next (asm reads issued one trip ahead, see tr_read_asm;
        // they are waited for by the lgkmcnt(0) of trip_sync)
        s16x4 dlo[2][DVB], dhi[2][DVB], qlo[2][DVB], qhi[2][DVB];
        auto prefetch = [&](auto stage_tag) {
            constexpr int QOFF = decltype(stage_tag)::value * STAGE, DOFF = QOFF + TILE;
            auto ld = [&](auto stag, auto dbtag) {
                constexpr int S_ = decltype(stag)::value, DB_ = decltype(dbtag)::value;
                if constexpr (DB_ < DVB) {
                    dlo[S_][DB_] = tr_read_asm<DOFF + 16 * S_ * ROWB>(ta1[DB_]);
                    dhi[S_][DB_] = tr_read_asm<DOFF + 16 * S_ * ROWB>(ta2[DB_]);
                    qlo[S_][DB_] = tr_read_asm<QOFF + 16 * S_ * ROWB>(ta1[DB_]);
                    qhi[S_][DB_] = tr_read_asm<QOFF + 16 * S_ * ROWB>(ta2[DB_]);
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            using I3 = std::integral_constant<int, 3>;
            ld(I0{}, I0{}); ld(I0{}, I1{}); ld(I0{}, I2{}); ld(I0{}, I3{});
            ld(I1{}, I0{}); ld(I1{}, I1{}); ld(I1{}, I2{}); ld(I1{}, I3{});
        };
        // trip t (t % 4 == K): DMA slice t+3 into stage (K+3)%4, accumulate slice t-1 (fragments fetched last trip),
        // then fetch slice t's transposed fragments from stage K for the next trip
        STAMP_DECL
        auto trip = [&](auto ktag, int t) {
            constexpr int K = decltype(ktag)::value;
            STAMP(7)
            if (t + 3 < n_it) stage_next(((K + 3) % NST) * STAGE);
            STAMP(0)
            if (t >= 1) {
                const char* x = xr + ((t - 1) & 1) * (4 * XBYTES);
                frag pP[2], pS[2];
                pP[0] = *reinterpret_cast<const frag*>(x);
                pP[1] = *reinterpret_cast<const frag*>(x + 1024);
                pS[0] = *reinterpret_cast<const frag*>(x + 2048);
                pS[1] = *reinterpret_cast<const frag*>(x + 3072);
                STAMP(1)
                if constexpr (SPILL) {
                    sv0 = __builtin_bit_cast(u32x4, pS[0]);
                    sv1 = __builtin_bit_cast(u32x4, pS[1]);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int db = 0; db < DVB; ++db) {
                        dVt[db] = M::run(__builtin_bit_cast(frag, join8(dlo[s][db], dhi[s][db])), pP[s], dVt[db]);
                        dKt[db] = M::run(__builtin_bit_cast(frag, join8(qlo[s][db], qhi[s][db])), pS[s], dKt[db]);
                    }
            }
            STAMP(3)
            if (t < n_it) {
                __builtin_amdgcn_sched_barrier(0);   // the asm reads overwrite the fragments: MFMAs above must be issued first
                prefetch(ktag);
            }
            if constexpr (SPILL) {
                if (t >= 1) {   // after the MFMAs and the read burst: the store issue hides under the LDS latency
                    __builtin_nontemporal_store(sv0, reinterpret_cast<u32x4*>(dsp));
                    __builtin_nontemporal_store(sv1, reinterpret_cast<u32x4*>(dsp + 1024));
                    dsp += kDsChunk;
                    if (++ac_q == nq) {
                        ac_q = 0;
                        dsp += ds_head_step;
                    }
                }
            }
            trip_sync(t + 3 < n_it, SPILL && t >= 1);
            STAMP(5)
        };
        int t = 0;
        for (; t + 4 <= n_trips; t += 4) {
            trip(std::integral_constant<int, 0>{}, t);
            trip(std::integral_constant<int, 1>{}, t + 1);
            trip(std::integral_constant<int, 2>{}, t + 2);
            trip(std::integral_constant<int, 3>{}, t + 3);
        }
        if (t < n_trips) trip(std::integral_constant<int, 0>{}, t++);
        if (t < n_trips) trip(std::integral_constant<int, 1>{}, t++);
        if (t < n_trips) trip(std::integral_constant<int, 2>{}, t++);
#ifdef SFA_STAMPS
        if (kb == 40 && hk == 0 && b == 0 && lane == 0) {
            for (int kk = 0; kk < 8; ++kk) g_stamps[wave * 8 + kk] = ts_acc_[kk];
            g_stamps[wave * 8 + 6] = (unsigned long long)n_it;
        }
#endif

        char* dkb = a.dk.ptr + ((int64_t)sq.bb * a.dk.sb + (int64_t)hk * a.dk.sh + (int64_t)sq.row0 * a.dk.sn) * 2;
        char* dvb = a.dv.ptr + ((int64_t)sq.bb * a.dv.sb + (int64_t)hk * a.dv.sh + (int64_t)sq.row0 * a.dv.sn) * 2;
        const __amdgpu_buffer_rsrc_t rdk = __builtin_amdgcn_make_buffer_rsrc((void*)dkb, 0, seq_range(a.cu, a.dk_range, Nk, a.dk.sn, D), 0x00020000);
        const __amdgpu_buffer_rsrc_t rdv = __builtin_amdgcn_make_buffer_rsrc((void*)dvb, 0, seq_range(a.cu, a.dv_range, Nk, a.dv.sn, D), 0x00020000);
        typedef __attribute__((ext_vector_type(4))) E e4;
#pragma unroll
        for (int db = 0; db < DVB; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * db + 8 * g4 + 4 * h;
                if (d < D) {
                    e4 pk, pv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pk[e] = (E)(dKt[db][4 * g4 + e] * a.scale);
                        pv[e] = (E)(dVt[db][4 * g4 + e]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rdk,
                                                          (unsigned)key * (unsigned)(a.dk.sn * 2) + (unsigned)(d * 2), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pv), rdv,
                                                          (unsigned)key * (unsigned)(a.dv.sn * 2) + (unsigned)(d * 2), 0, 0);
                }
            }
    }
}

// ========================================================================= dQ
template <typename T, int D, int NW>
__global__ __launch_bounds__(NW * 64, 2) void bwd_dq_mfma_kernel(BwdArgs a) {
    using M = Mma<T>;
    using frag = typename M::frag;
    using E = typename M::elem;
    constexpr int DK = D / 16;
    constexpr int DVB = (D + 31) / 32;
    constexpr int CPR = D / 8;
    constexpr int ROWB = (D <= 64) ? 128 : 256;
    constexpr int TILE_BYTES = 64 * ROWB;
    constexpr int NT = NW * 64;
    constexpr int NCH = 64 * CPR;
    constexpr int NLD = (NCH + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = a.n_qtiles - 1 - bid % a.n_qtiles;      // longest (latest) q tiles first, as in the forward
    int rest = bid / a.n_qtiles;
    const int hg = rest % a.hgroups;
    rest /= a.hgroups;
    const int hk = rest % a.Hkv;
    const int b = rest / a.Hkv;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int g = a.Hq / a.Hkv;
    const int hh = wave % a.hpw, rbi = wave / a.hpw;
    const int head = hk * g + hg * a.hpw + hh;
    const SeqInfo sq = seq_of(a.cu, b, a.N);
    const int N = sq.N, ns = a.num_sink;
    const int P = a.cu ? 0 : a.Nk - a.N;         // position of query row 0 among the keys
    const int W = a.window < N + P ? a.window : N + P;
    const int BM = 32 * a.rb;
    const int q0 = qt * BM;
    if (q0 >= N) return;   // packed batches: the grid is sized for the longest sequence
    const int q1 = (q0 + BM < N) ? q0 + BM : N;
    const int qw0 = q0 + 32 * rbi;
    const int qw_hi = (qw0 + 31 < N - 1) ? qw0 + 31 : N - 1;
    const int qrow = qw0 + r;
    const bool wave_live = qw0 < N;
    const int pw0 = qw0 + P, pw_hi = qw_hi + P;  // the wave's rows as key positions

    const int ns_eff = ns < q1 + P ? ns : q1 + P;
    const int ts_hi = (ns_eff + 63) >> 6;
    int wlo = q0 + P - W + 1;
    if (wlo < 0) wlo = 0;
    int tw_lo = wlo >> 6;
    if (tw_lo < ts_hi) tw_lo = ts_hi;
    const int tw_hi = (q1 + P + 63) >> 6;
    if (tw_lo > tw_hi) tw_lo = tw_hi;
    const int nt = ts_hi + (tw_hi - tw_lo);

    const char* qb = a.q.ptr + ((int64_t)sq.bb * a.q.sb + (int64_t)head * a.q.sh + (int64_t)sq.row0 * a.q.sn) * 2;
    const char* dob = a.d_o.ptr + ((int64_t)sq.bb * a.d_o.sb + (int64_t)head * a.d_o.sh + (int64_t)sq.row0 * a.d_o.sn) * 2;
    const char* kb = a.k.ptr + ((int64_t)sq.bb * a.k.sb + (int64_t)hk * a.k.sh + (int64_t)sq.row0 * a.k.sn) * 2;
    const char* vb = a.v.ptr + ((int64_t)sq.bb * a.v.sb + (int64_t)hk * a.v.sh + (int64_t)sq.row0 * a.v.sn) * 2;
    char* dqb = a.dq.ptr + ((int64_t)sq.bb * a.dq.sb + (int64_t)head * a.dq.sh + (int64_t)sq.row0 * a.dq.sn) * 2;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)qb, 0, seq_range(a.cu, a.q_range, N, a.q.sn, D), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdo = __builtin_amdgcn_make_buffer_rsrc((void*)dob, 0, seq_range(a.cu, a.do_range, N, a.d_o.sn, D), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, seq_range(a.cu, a.k_range, N, a.k.sn, D), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, seq_range(a.cu, a.v_range, N, a.v.sn, D), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdq = __builtin_amdgcn_make_buffer_rsrc((void*)dqb, 0, seq_range(a.cu, a.dq_range, N, a.dq.sn, D), 0x00020000);

    frag qf[DK], dof[DK];
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) {
        const unsigned qo = (unsigned)qrow * (unsigned)(a.q.sn * 2) + (unsigned)((2 * ks + h) * 16);
        const unsigned oo = (unsigned)qrow * (unsigned)(a.d_o.sn * 2) + (unsigned)((2 * ks + h) * 16);
        qf[ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rq, qo, 0, 0));
        dof[ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rdo, oo, 0, 0));
    }
    float lse2 = INFINITY, dlt = 0.f;   // rows >= N: p = exp2(-inf) = 0
    if (qrow < N) {
        const int64_t idx = (a.cu ? (int64_t)head * a.n_total + sq.row0 : ((int64_t)b * a.Hq + head) * N) + qrow;
        lse2 = a.lse[idx] * kLog2e;
        dlt = a.delta[idx];
    }

    u32x4 kst[NLD], vst[NLD];
    const unsigned ksn2 = (unsigned)(a.k.sn * 2), vsn2 = (unsigned)(a.v.sn * 2);
    auto issue_loads = [&](int t) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NT;
            const int key = c / CPR, ch = c % CPR;
            const bool in = (NCH % NT == 0) || (c < NCH);
            const unsigned row = (unsigned)(64 * t + key);
            const unsigned ko = in ? row * ksn2 + (unsigned)(ch * 16) : 0xFFFFFFF0u;
            const unsigned vo = in ? row * vsn2 + (unsigned)(ch * 16) : 0xFFFFFFF0u;
            kst[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, ko, 0, 0);
            vst[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, vo, 0, 0);
        }
    };
    auto write_lds = [&](int buf) {
        char* kl = smem + buf * 2 * TILE_BYTES;
        char* vl = kl + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NT;
            const int key = c / CPR, ch = c % CPR;
            if ((NCH % NT == 0) || (c < NCH)) {
                const int o = key * ROWB + ((ch ^ sw<ROWB>(key)) << 4);
                *reinterpret_cast<u32x4*>(kl + o) = kst[i];
                *reinterpret_cast<u32x4*>(vl + o) = vst[i];
            }
        }
    };
    auto tile_of = [&](int it) { return it < ts_hi ? it : tw_lo + (it - ts_hi); };

    const int rowrd = r * ROWB;
    const int rsw = sw<ROWB>(r);               // sw(32*kh + r) == sw(r)
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const int tr_row = 4 * h + q4;
    const int tr_col = 2 * g1 + (p4 >> 1);
    const int tr_byte = (p4 & 1) * 8;
    const float c = a.scale_log2;

    f32x16 dQt[DVB];
#pragma unroll
    for (int db = 0; db < DVB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) dQt[db][i] = 0.f;

    if (nt > 0) {
        issue_loads(tile_of(0));
        write_lds(0);
    }
    __syncthreads();

    for (int it = 0; it < nt; ++it) {
        const int buf = it & 1;
        if (it + 1 < nt) issue_loads(tile_of(it + 1));
        const int k0 = tile_of(it) * 64;
        const bool needed = wave_live && (k0 <= pw_hi) && (k0 < ns || k0 + 63 >= pw0 - W + 1);
        if (needed) {
            const bool full = (k0 + 63 <= pw0) && ((k0 + 63 < ns) || (k0 >= pw_hi - W + 1));
            const char* kl = smem + buf * 2 * TILE_BYTES;
            const char* vl = kl + TILE_BYTES;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                f32x16 S, dP;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    S[i] = 0.f;
                    dP[i] = 0.f;
                }
                // (batching the K/V fragment reads ahead of the chains, as the forward does, costs 26 more VGPRs here
                // and measured slower: 2.47 vs 2.33 ms at C3)
#pragma unroll
                for (int ks = 0; ks < DK; ++ks) {
                    const int o = kh * 32 * ROWB + rowrd + (((2 * ks + h) ^ rsw) << 4);
                    const frag kf = *reinterpret_cast<const frag*>(kl + o);
                    const frag vf = *reinterpret_cast<const frag*>(vl + o);
                    S = M::run(kf, qf[ks], S);
                    dP = M::run(vf, dof[ks], dP);
                }
                frag pS[2];
                if (full) {
                    // packed f32 math (two elements per VALU issue slot) around the 16 exp2
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 c2 = {c, c}, nl2 = {-lse2, -lse2}, nd2 = {-dlt, -dlt};
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {
                        const f32x2 sv = {S[i], S[i + 1]}, dv = {dP[i], dP[i + 1]};
                        const f32x2 x = __builtin_elementwise_fma(sv, c2, nl2);
                        f32x2 p;
                        p[0] = __builtin_amdgcn_exp2f(x[0]);
                        p[1] = __builtin_amdgcn_exp2f(x[1]);
                        const f32x2 ds = p * (dv + nd2);
                        pS[i >> 3][i & 7] = (E)ds[0];
                        pS[i >> 3][(i & 7) + 1] = (E)ds[1];
                    }
                } else {
                    int qm = qrow + P;
                    asm volatile("; edge tile" : "+v"(qm) : : "memory");     // side effect: the branch cannot be speculated / if-converted
                    // as in the forward: most edge tiles need only the causal OR only the window test, one compare
                    // of the element's constant offset against a per-lane threshold
                    // (head dims below 128 only: at D = 128 the extra paths push the kernel over its register budget)
                    const bool no_sink = D < 128 && k0 >= ns;
                    if (no_sink && k0 >= pw_hi - W + 1) {
                        const int t = qm - k0 - 32 * kh - 4 * h;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float p = __builtin_amdgcn_exp2f(fmaf(S[i], c, -lse2));
                            pS[i >> 3][i & 7] = (E)(((i & 3) + 8 * (i >> 2) <= t) ? p * (dP[i] - dlt) : 0.f);
                        }
                    } else if (no_sink && k0 + 63 <= pw0) {
                        const int t = qm - W - k0 - 32 * kh - 4 * h;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float p = __builtin_amdgcn_exp2f(fmaf(S[i], c, -lse2));
                            pS[i >> 3][i & 7] = (E)(((i & 3) + 8 * (i >> 2) > t) ? p * (dP[i] - dlt) : 0.f);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int key = k0 + 32 * kh + (i & 3) + 8 * (i >> 2) + 4 * h;
                            const bool valid = (key <= qm) && (key < ns || key + W > qm);
                            const float p = __builtin_amdgcn_exp2f(fmaf(S[i], c, -lse2));
                            pS[i >> 3][i & 7] = (E)(valid ? p * (dP[i] - dlt) : 0.f);
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int db = 0; db < DVB; ++db) {
                        const int row = 32 * kh + 16 * s + tr_row;
                        const int o1 = row * ROWB + (((4 * db + tr_col) ^ sw<ROWB>(row)) << 4) + tr_byte;
                        const int o2 = (row + 8) * ROWB + (((4 * db + tr_col) ^ sw<ROWB>(row + 8)) << 4) + tr_byte;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(kl + o1));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(kl + o2));
                        s16x8 vv;
                        vv[0] = lo[0]; vv[1] = lo[1]; vv[2] = lo[2]; vv[3] = lo[3];
                        vv[4] = hi[0]; vv[5] = hi[1]; vv[6] = hi[2]; vv[7] = hi[3];
                        dQt[db] = M::run(__builtin_bit_cast(frag, vv), pS[s], dQt[db]);
                    }
            }
        }
        if (it + 1 < nt) write_lds(buf ^ 1);
        __syncthreads();
    }

    // dQ = scale * dQ^T^T (sink_flash_attention.py:481); 16-byte stores after one v_permlane32_swap per register (the
    // two half-waves hold the two 8-byte halves of every 16-byte piece of a row, see the forward's store tail)
    const unsigned orow = (unsigned)qrow * (unsigned)(a.dq.sn * 2);
    typedef __attribute__((ext_vector_type(4))) E e4;
#pragma unroll
    for (int db = 0; db < DVB; ++db)
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            if (32 * db + 16 * p2 < D) {
                e4 pe, po;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pe[e] = (E)(dQt[db][8 * p2 + e] * a.scale);
                    po[e] = (E)(dQt[db][8 * p2 + 4 + e] * a.scale);
                }
                const u32x2 ev = __builtin_bit_cast(u32x2, pe), od = __builtin_bit_cast(u32x2, po);
                const auto s0 = __builtin_amdgcn_permlane32_swap(ev[0], od[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(ev[1], od[1], false, false);
                u32x4 w;
                w[0] = s0[0]; w[1] = s1[0]; w[2] = s0[1]; w[3] = s1[1];
                const int d = 32 * db + 16 * p2 + 8 * h;
                const unsigned off = wave_live ? orow + (unsigned)(d * 2) : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_buffer_store_b128(w, rdq, off, 0, 0);
            }
        }
}

// =============================================================== dQ from spilled dS (plain GEMM, HBM-bound)
// dQ^T[d, i] = sum_j K^T[d, j] dS^T[j, i] over the dS chunks the dK/dV kernel saved: no S / dP recompute.
// Workgroup = 8 compute waves = 8 consecutive 32-row slices of one (b, q head) + 1 LOADER wave.  A stage = 64 keys.
//   * The K half tile is shared: the loader wave brings it by LDS-DMA into a KR-deep ring (swizzled image, K^T
//     fragments by transposed reads) one stage ahead and signals it with the workgroup barrier.
//   * Each compute wave streams ITS OWN dS (4 KB per stage) into a private R-deep ring, R-1 stages in flight, gated
//     only by its own counted vmcnt (vmcnt retires in order, so the deep dS stream must not share a wave with the
//     shallow K stream: that is why K has its own wave).  The stream from HBM (2 bytes per (query, key) pair and
//     head) is what bounds the kernel; ~100 KB in flight per CU cover the ~3.5 us loaded latency.
//   * The dS image in LDS is [key][64 B], gathered by the DMA from the stored fragments ([key group][sigma = 2 s2 +
//     h][key][16 B], written by the dK/dV kernel with 1 KB-contiguous stores): a transposed read turns it into the
//     B operand, its N index being the STORED row order, undone at the dQ store.
//   * A slice that a key block does not cover has no chunk: its offset falls outside the block's descriptor range
//     and the DMA zero-fills.
template <int N>
using IC = std::integral_constant<int, N>;

template <typename T, int D, int R, int KR>
__global__ __launch_bounds__(576, 1) void bwd_dq_gemm_kernel(BwdArgs a) {
    using M = Mma<T>;
    using frag = typename M::frag;
    using E = typename M::elem;
    constexpr int NW = 8;
    constexpr int DVB = (D + 31) / 32;
    constexpr int CPR = D / 8;
    constexpr int ROWB = (D <= 64) ? 128 : 256;
    constexpr int SPR = ROWB / 16;                   // 16-byte slots per K row
    constexpr int KBYTES = 64 * ROWB;                // K half tile
    constexpr int DSOFF = KR * KBYTES;               // dS rings start here: [wave][slot][4096]
    constexpr int NDK = (64 * SPR) / 64;             // K DMA instructions per stage (loader wave)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nqb = (a.N + 32 * NW - 1) / (32 * NW);
    const int qb = bid % nqb;
    const int bh = bid / nqb;
    const int head = bh % a.Hq, b = bh / a.Hq;
    const int g = a.Hq / a.Hkv;
    const int hk = head / g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int N = a.N, W = a.window, ns = a.num_sink;
    const int q0 = qb * 32 * NW;                       // first row of the workgroup
    const int q_last = (q0 + 32 * NW < N ? q0 + 32 * NW : N) - 1;

    // key blocks to visit: sink blocks [0, nsb) then window blocks [kb_lo, kb_hi]
    const int kb_hi = q_last / kKB;
    int nsb = (ns + kKB - 1) / kKB;
    if (nsb > kb_hi + 1) nsb = kb_hi + 1;
    int wlo = q0 - W + 1;
    if (wlo < 0) wlo = 0;
    int kb_lo = wlo / kKB;
    if (kb_lo < nsb) kb_lo = nsb;
    const int n_blocks = nsb + (kb_hi >= kb_lo ? kb_hi - kb_lo + 1 : 0);
    const int n_st = 2 * n_blocks;                     // stages (64 keys each)
    auto block_of = [&](int i) { return i < nsb ? i : kb_lo + (i - nsb); };

    if (wave == NW) {
        // ------------------------------------------------------------------ loader wave: the K ring
        const char* kbase = a.k.ptr + ((int64_t)b * a.k.sb + (int64_t)hk * a.k.sh) * 2;
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, a.k_range, 0x00020000);
        const unsigned ksn2 = (unsigned)(a.k.sn * 2);
        unsigned vk0[NDK];     // LDS slot c = 64 i + lane -> (row, slot); source chunk = slot ^ sw(row)
#pragma unroll
        for (int i = 0; i < NDK; ++i) {
            const int cidx = i * 64 + lane;
            const int row = cidx / SPR;
            const int ch = (cidx % SPR) ^ sw<ROWB>(row);
            vk0[i] = ch < CPR ? (unsigned)row * ksn2 + (unsigned)(ch * 16) : 0xFFFFFFF0u;
        }
        auto issue_k = [&](int st, int buf_off) {
            const unsigned key0 = (unsigned)(block_of(st >> 1) * kKB + (st & 1) * 64);
#pragma unroll
            for (int i = 0; i < NDK; ++i) {
                const unsigned ko = vk0[i] == 0xFFFFFFF0u ? vk0[i] : vk0[i] + key0 * ksn2;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)(smem + buf_off + i * 1024),
                                                         16, ko, 0, 0, 0);
            }
        };
        if (n_st > 0) issue_k(0, 0);
        int kbuf = 0;
        for (int st = 0; st < n_st; ++st) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // K(st) landed
            __builtin_amdgcn_s_barrier();                          // ... and every reader of K(st + 1 - KR) is done
            __builtin_amdgcn_sched_barrier(0);
            kbuf = kbuf + 1 == KR ? 0 : kbuf + 1;
            if (st + 1 < n_st) issue_k(st + 1, kbuf * KBYTES);
        }
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    const int qt = qb * NW + wave;                     // this wave's slice
    const char* dsh = a.ds + ((int64_t)b * a.Hq + head) * a.ds_chunks * kDsChunk;
    char* ring = smem + DSOFF + wave * (R * 4096);
    int pf_next = n_st > 0 ? a.ds_prefix[block_of(0)] : 0;
    int nq_next = n_st > 0 ? a.ds_prefix[block_of(0) + 1] - pf_next : 0;
    const unsigned so_lane = (unsigned)((lane & 3) * 512 + (lane >> 2) * 16);
    auto issue_ds = [&](int st, int slot_off) {
        const int kb = block_of(st >> 1), half = st & 1;
        // descriptor = the block's chunk range of this head, so a slice the block does not cover reads zeros
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(dsh + (int64_t)pf_next * kDsChunk), 0, (unsigned)nq_next * (unsigned)kDsChunk, 0x00020000);
        // LDS slot c = 64 i + lane -> key 16 (i & 1) + lane / 4 of key group i / 2, sigma = lane % 4
        const unsigned so = (unsigned)(qt - 4 * kb) * (unsigned)kDsChunk + (unsigned)(half * 4096) + so_lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            auto dst = (__attribute__((address_space(3))) void*)(ring + slot_off + i * 1024);
            const unsigned o = so + (unsigned)((i >> 1) * 2048 + (i & 1) * 256);
            if (a.ds_nt)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, dst, 16, o, 0, 0, 2);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, dst, 16, o, 0, 0, 0);
        }
        if (half == 1 && st + 1 < n_st) {              // next block's table entries, one stage ahead of their use
            const int kn = block_of((st + 1) >> 1);
            pf_next = a.ds_prefix[kn];
            nq_next = a.ds_prefix[kn + 1] - pf_next;
        }
    };

    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const int tr_row = 4 * h + q4;
    const int tr_col = 2 * g1 + (p4 >> 1);
    const int tr_byte = (p4 & 1) * 8;
    const unsigned smem_base = (unsigned)(size_t)smem;
    unsigned ta1[DVB], ta2[DVB];
#pragma unroll
    for (int db = 0; db < DVB; ++db) {
        ta1[db] = smem_base + (unsigned)(tr_row * ROWB + (((4 * db + tr_col) ^ sw<ROWB>(tr_row)) << 4) + tr_byte);
        ta2[db] = smem_base + (unsigned)((tr_row + 8) * ROWB + (((4 * db + tr_col) ^ sw<ROWB>(tr_row + 8)) << 4) + tr_byte);
    }
    const unsigned tb = smem_base + (unsigned)(DSOFF + wave * (R * 4096) + tr_row * 64 + tr_col * 16 + tr_byte);

    f32x16 dQt[DVB];
#pragma unroll
    for (int db = 0; db < DVB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) dQt[db][i] = 0.f;

#pragma unroll
    for (int i = 0; i < R - 1; ++i)
        if (i < n_st) issue_ds(i, i * 4096);

    // stage st (st % R == K): own dS(st) landed (counted), barrier (K(st) landed), refill the ring slot freed by
    // stage st-1, transposed reads, 16 MFMAs
    auto stage = [&](auto ktag, auto kbtag, int st) {
        constexpr int K = decltype(ktag)::value;
        constexpr unsigned KOFF = decltype(kbtag)::value * KBYTES;
        constexpr unsigned SOFF = K * 4096;
        const int ahead = n_st - 1 - st < R - 2 ? n_st - 1 - st : R - 2;   // own dS stages younger than st in flight
        if (ahead >= 2)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (st + R - 1 < n_st) issue_ds(st + R - 1, ((K + R - 1) % R) * 4096);
        s16x4 blo[2][2], bhi[2][2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            blo[kh][0] = tr_read_asm<0>(tb + SOFF + kh * 2048);
            bhi[kh][0] = tr_read_asm<512>(tb + SOFF + kh * 2048);
            blo[kh][1] = tr_read_asm<1024>(tb + SOFF + kh * 2048);
            bhi[kh][1] = tr_read_asm<1536>(tb + SOFF + kh * 2048);
        }
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            s16x4 alo[2][DVB], ahi[2][DVB];
#pragma unroll
            for (int db = 0; db < DVB; ++db) {
                alo[0][db] = tr_read_asm<0>(ta1[db] + KOFF + kh * 32 * ROWB);
                ahi[0][db] = tr_read_asm<0>(ta2[db] + KOFF + kh * 32 * ROWB);
                alo[1][db] = tr_read_asm<16 * ROWB>(ta1[db] + KOFF + kh * 32 * ROWB);
                ahi[1][db] = tr_read_asm<16 * ROWB>(ta2[db] + KOFF + kh * 32 * ROWB);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int db = 0; db < DVB; ++db)
                    dQt[db] = M::run(__builtin_bit_cast(frag, join8(alo[s2][db], ahi[s2][db])),
                                     __builtin_bit_cast(frag, join8(blo[kh][s2], bhi[kh][s2])), dQt[db]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_assert(R == 4 || R == 3, "ring depth");
    static_assert(R % KR == 0, "the unrolled stage index must determine the K buffer");
    int st = 0;
    if constexpr (R == 4) {
        for (; st + 4 <= n_st; st += 4) {
            stage(IC<0>{}, IC<0 % KR>{}, st);
            stage(IC<1>{}, IC<1 % KR>{}, st + 1);
            stage(IC<2>{}, IC<2 % KR>{}, st + 2);
            stage(IC<3>{}, IC<3 % KR>{}, st + 3);
        }
        if (st < n_st) stage(IC<0>{}, IC<0 % KR>{}, st++);
        if (st < n_st) stage(IC<1>{}, IC<1 % KR>{}, st++);
        if (st < n_st) stage(IC<2>{}, IC<2 % KR>{}, st++);
    } else {
        for (; st + 3 <= n_st; st += 3) {
            stage(IC<0>{}, IC<0 % KR>{}, st);
            stage(IC<1>{}, IC<1 % KR>{}, st + 1);
            stage(IC<2>{}, IC<2 % KR>{}, st + 2);
        }
        if (st < n_st) stage(IC<0>{}, IC<0 % KR>{}, st++);
        if (st < n_st) stage(IC<1>{}, IC<1 % KR>{}, st++);
    }

    // dQ = scale * dQ^T^T; accumulator column n = 8*sigma + pos = 16 s2 + 8 h' + 4 qb + e holds row
    // 16 s2 + 8 qb + 4 h' + e of the slice (bits 2 and 3 of n swapped)
    const int n = lane & 31;
    const int qrow = qt * 32 + (n & 19) + ((n & 4) << 1) + ((n & 8) >> 1);
    char* dqb = a.dq.ptr + ((int64_t)b * a.dq.sb + (int64_t)head * a.dq.sh) * 2;
    const __amdgpu_buffer_rsrc_t rdq = __builtin_amdgcn_make_buffer_rsrc((void*)dqb, 0, a.dq_range, 0x00020000);
    const unsigned orow = (unsigned)qrow * (unsigned)(a.dq.sn * 2);
    typedef __attribute__((ext_vector_type(4))) E e4;
#pragma unroll
    for (int db = 0; db < DVB; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = 32 * db + 8 * g4 + 4 * h;
            if (d < D) {
                e4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (E)(dQt[db][4 * g4 + e] * a.scale);
                const unsigned off = qrow < N ? orow + (unsigned)(d * 2) : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rdq, off, 0, 0);
            }
        }
}

// exclusive prefix of dkdv_slices() over the key blocks: the chunk index table of the dS spill
__global__ __launch_bounds__(256) void ds_prefix_kernel(int* __restrict__ tbl, int n_kb, int N, int ns, int W) {
    __shared__ int part[256];
    const int t = threadIdx.x;
    const int per = (n_kb + 255) / 256;
    const int lo = t * per < n_kb ? t * per : n_kb;
    const int hi = lo + per < n_kb ? lo + per : n_kb;
    int sum = 0;
    for (int kb = lo; kb < hi; ++kb) sum += dkdv_slices(kb, N, ns, W);
    part[t] = sum;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) {
            const int x = part[i];
            part[i] = run;
            run += x;
        }
    }
    __syncthreads();
    int run = part[t];
    for (int kb = lo; kb < hi; ++kb) {
        tbl[kb] = run;
        run += dkdv_slices(kb, N, ns, W);
    }
    if (hi == n_kb && (lo < hi || t == 0)) tbl[n_kb] = run;
}

// row constants for the wave-specialised dK/dV kernel: consts[b,h,0,:] = -LSE*log2e, consts[b,h,1,:] = -Delta
__global__ __launch_bounds__(256) void bwd_consts_kernel(const float* __restrict__ lse, const float* __restrict__ delta,
                                                        float* __restrict__ consts, int N, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t bh = i / N;
    const int n = (int)(i - bh * N);
    consts[(bh * 2) * N + n] = -lse[i] * kLog2e;
    consts[(bh * 2 + 1) * N + n] = -delta[i];
}

// ==================================================================== launch
int dkdv_mode() {
    static const int mode = [] {
        const char* e = getenv("SFA_DKDV");
        return e ? atoi(e) : 3;
    }();
    return mode;
}

// dS spill geometry: chunks per head (0 = spill not possible for this problem)
int64_t ds_chunks_per_head(const Problem& p) {
    const int W = p.window < 0 ? 0 : (p.window > p.N ? p.N : p.window);
    const int n_kb = (int)cdiv64(p.N, kKB);
    int64_t total = 0;
    for (int kb = 0; kb < n_kb; ++kb) total += dkdv_slices(kb, p.N, p.num_sink, W);
    if (total * kDsChunk >= (1ll << 31)) return 0;   // 32-bit offsets inside one head's chunk range
    return total;
}
size_t ds_table_bytes(const Problem& p) { return (((size_t)cdiv64(p.N, kKB) + 1) * sizeof(int) + 255) & ~(size_t)255; }

template <typename T, int D>
int launch_bwd(const BwdArgs& a, int B, hipStream_t stream) {
    constexpr int ROWB = (D <= 64) ? 128 : 256;
    // dK/dV variant (SFA_DKDV; C3, ms): 3 = wave-specialised score/accumulate waves, 2 per SIMD (default, 3.6);
    // anything else = the plain kernel, 4 waves, one per SIMD (5.0).  Two more variants were measured and removed:
    // 8 identical waves with K/V fragments from LDS (LDS-bound, 7.4) and a software-pipelined 4-wave kernel
    // (register-starved under hipcc 7.2, 6.8).
    const int mode = dkdv_mode();
    if (mode == 3) {
        if (!a.consts_ready) {
            const int rows = a.cu ? a.n_total : a.N;                  // packed batches: one [Hq, n_total] slab
            const int64_t total = (a.cu ? 1 : (int64_t)B) * a.Hq * rows;
            bwd_consts_kernel<<<dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, stream>>>(
                a.lse, a.delta, const_cast<float*>(a.consts), rows, total);
        }
        constexpr int lds = 4 * (2 * 32 * ROWB + 256) + 2 * 4 * 4096;
        const int nblk = a.n_kblocks * a.Hkv * B;
        if (a.ds) {
            // dS spill: chunk table, dK/dV kernel that also saves dS, then dQ as a GEMM over the saved dS
            ds_prefix_kernel<<<dim3(1), dim3(256), 0, stream>>>(const_cast<int*>(a.ds_prefix), a.n_kblocks, a.N,
                                                                a.num_sink, a.window);
            auto kern = bwd_dkdv_mfma_kernel_ws<T, D, true>;
            static unsigned long long done = 0;
            ensure_dynamic_lds((const void*)kern, lds, &done);
            kern<<<dim3(nblk), dim3(512), lds, stream>>>(a);
            int st = launch_status("bwd_dkdv_mfma_ws_spill");
            if (st) return st;
            record_stage(2, stream);
            static const int gr = [] {      // dS ring depth of the dQ GEMM: 4 (160 KB LDS) or 3 (144 KB)
                const char* e = getenv("SFA_DQ_GEMM_R");
                return e && atoi(e) == 3 ? 3 : 4;
            }();
            const int64_t gblk = (int64_t)B * a.Hq * cdiv64(a.N, 32 * 8);
            if (gr == 4) {
                constexpr int glds = 2 * 64 * ROWB + 8 * 4 * 4096;
                auto gk = bwd_dq_gemm_kernel<T, D, 4, 2>;
                static unsigned long long gdone = 0;
                ensure_dynamic_lds((const void*)gk, glds, &gdone);
                gk<<<dim3((unsigned)gblk), dim3(576), glds, stream>>>(a);
            } else {
                constexpr int glds = 3 * 64 * ROWB + 8 * 3 * 4096;
                auto gk = bwd_dq_gemm_kernel<T, D, 3, 3>;
                static unsigned long long gdone = 0;
                ensure_dynamic_lds((const void*)gk, glds, &gdone);
                gk<<<dim3((unsigned)gblk), dim3(576), glds, stream>>>(a);
            }
            const int gnw = gr;
            set_path("bwd_mfma_%s_d%d_dkdvws8_spill_dqgemm_r%d", DT<T>::id == SFA_DTYPE_BF16 ? "bf16" : "f16", D, gnw);
            return launch_status("bwd_dq_gemm");
        }
        auto kern = bwd_dkdv_mfma_kernel_ws<T, D, false>;
        static unsigned long long done = 0;
        ensure_dynamic_lds((const void*)kern, lds, &done);
        kern<<<dim3(nblk), dim3(512), lds, stream>>>(a);
        int st = launch_status("bwd_dkdv_mfma_ws");
        if (st) return st;
    } else {
        constexpr int lds = 2 * (2 * 64 * ROWB + 512);
        auto kern = bwd_dkdv_mfma_kernel<T, D>;
        static unsigned long long done = 0;
        ensure_dynamic_lds((const void*)kern, lds, &done);
        const int nblk = a.n_kblocks * a.Hkv * B;
        kern<<<dim3(nblk), dim3(256), lds, stream>>>(a);
        int st = launch_status("bwd_dkdv_mfma");
        if (st) return st;
    }
    record_stage(2, stream);
    // dQ kernel: forward-shaped, same choice of workgroup size as the forward (short window -> two 4-wave workgroups
    // per CU instead of one 8-wave workgroup)
    static const int dq_nw_env = [] {
        const char* e = getenv("SFA_DQ_NW");
        return e ? atoi(e) : 0;
    }();
    const int dq_nw = (dq_nw_env == 4 || dq_nw_env == 8) ? dq_nw_env : (a.window <= 2048 ? 4 : 8);
    if (dq_nw == 4) {
        constexpr int NW = 4;
        constexpr int lds = 2 * 2 * 64 * ROWB;
        BwdArgs a4 = a;
        const int g = a.Hq / a.Hkv;
        a4.hpw = g % 4 == 0 ? 4 : (g % 2 == 0 ? 2 : 1);
        a4.rb = NW / a4.hpw;
        a4.n_qtiles = (int)cdiv64(a.N, 32 * a4.rb);
        a4.hgroups = g / a4.hpw;
        auto kern = bwd_dq_mfma_kernel<T, D, NW>;
        static unsigned long long done = 0;
        ensure_dynamic_lds((const void*)kern, lds, &done);
        const int64_t nblk = (int64_t)a4.n_qtiles * a4.hgroups * a.Hkv * B;
        kern<<<dim3((unsigned)nblk), dim3(NW * 64), lds, stream>>>(a4);
    } else {
        constexpr int NW = 8;
        constexpr int lds = 2 * 2 * 64 * ROWB;
        auto kern = bwd_dq_mfma_kernel<T, D, NW>;
        static unsigned long long done = 0;
        ensure_dynamic_lds((const void*)kern, lds, &done);
        const int nblk = a.n_qtiles * a.hgroups * a.Hkv * B;
        kern<<<dim3(nblk), dim3(NW * 64), lds, stream>>>(a);
    }
    set_path("bwd_mfma_%s_d%d_dkdv%s_dq%dw_hpw%d", DT<T>::id == SFA_DTYPE_BF16 ? "bf16" : "f16", D, mode == 3 ? "ws8" : "4w", dq_nw, a.hpw);
    return launch_status("bwd_dq_mfma");
}

template <typename T>
int launch_bwd_d(const BwdArgs& a, int D, int B, hipStream_t stream) {
    switch (D) {
        case 64: return launch_bwd<T, 64>(a, B, stream);
        case 80: return launch_bwd<T, 80>(a, B, stream);
        case 96: return launch_bwd<T, 96>(a, B, stream);
        case 128: return launch_bwd<T, 128>(a, B, stream);
    }
    set_error("bwd_mfma: unsupported head dim %d", D);
    return SFA_ERR_UNSUPPORTED;
}

int gcd(int x, int y) { return y == 0 ? x : gcd(y, x % y); }

unsigned slice_range(const sfa_tensor* t) { return (unsigned)(((t->shape[2] - 1) * t->stride[2] + t->shape[3]) * 2); }
bool slice_ok(const sfa_tensor* t) {
    const int64_t reach = (t->shape[2] + 320) * t->stride[2] * 2 + 512;
    return reach < (1ll << 32) - 65536 && ((uintptr_t)t->ptr % 16) == 0 && (t->stride[0] * 2) % 16 == 0 &&
           (t->stride[1] * 2) % 16 == 0 && (t->stride[2] * 2) % 16 == 0;
}

}  // namespace

bool bwd_mfma_supported(int dtype, int D) {
    return (dtype == SFA_DTYPE_BF16 || dtype == SFA_DTYPE_F16) && (D == 64 || D == 80 || D == 96 || D == 128);
}

static size_t consts_bytes(const Problem& p) {
    const size_t rows = p.cu ? (size_t)p.n_total : (size_t)p.B * p.N;
    return (rows * p.Hq * 2 * sizeof(float) + 255) & ~(size_t)255;
}

bool bwd_mfma_spill(const Problem& p, unsigned flags) {
    return (flags & SFA_FLAG_BWD_SPILL_DS) && dkdv_mode() == 3 && p.cu == nullptr && (p.Nk == 0 || p.Nk == p.N) &&
           ds_chunks_per_head(p) > 0;
}

// packed batches are served by the default (wave-specialised) dK/dV kernel only
bool bwd_mfma_varlen_ok() { return dkdv_mode() == 3; }
bool bwd_mfma_wants_consts() { return dkdv_mode() == 3; }

size_t bwd_mfma_workspace_bytes(const Problem& p, int, unsigned flags) {
    size_t n = consts_bytes(p);
    if (bwd_mfma_spill(p, flags))
        n += ds_table_bytes(p) + (size_t)p.B * p.Hq * (size_t)ds_chunks_per_head(p) * kDsChunk;
    return n;
}

int bwd_mfma(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* d_o, const float* lse,
             const float* delta, const sfa_tensor* dq, const sfa_tensor* dk, const sfa_tensor* dv, void* workspace,
             const Problem& p, unsigned flags, hipStream_t stream, bool consts_ready) {
    if ((p.cu || (p.Nk > 0 && p.Nk != p.N)) && !bwd_mfma_varlen_ok()) {
        set_error("packed (varlen) / N_q != N_kv backward is served by the default dK/dV kernel only (SFA_DKDV=3)");
        return SFA_ERR_UNSUPPORTED;
    }
    if (!(slice_ok(q) && slice_ok(k) && slice_ok(v) && slice_ok(d_o) && slice_ok(dq) && slice_ok(dk) && slice_ok(dv))) {
        if (p.cu || (p.Nk > 0 && p.Nk != p.N)) {
            set_error("packed (varlen) / N_q != N_kv backward needs 16-byte aligned rows and < 4 GiB head slices");
            return SFA_ERR_UNSUPPORTED;
        }
        return bwd_generic(q, k, v, d_o, lse, delta, dq, dk, dv, p, stream);
    }
    const int g = p.Hq / p.Hkv;
    BwdArgs a;
    a.q = make_view(q); a.k = make_view(k); a.v = make_view(v); a.d_o = make_view(d_o);
    a.dq = make_view(dq); a.dk = make_view(dk); a.dv = make_view(dv);
    a.lse = lse; a.delta = delta;
    a.consts = reinterpret_cast<const float*>(workspace);
    a.consts_ready = consts_ready ? 1 : 0;
    a.cu = p.cu;
    a.n_total = p.n_total;
    a.Nk = p.cu ? p.N : (p.Nk > 0 ? p.Nk : p.N);
    a.ds = nullptr;
    a.ds_prefix = nullptr;
    a.ds_chunks = 0;
    if (bwd_mfma_spill(p, flags)) {
        a.ds_prefix = reinterpret_cast<const int*>((char*)workspace + consts_bytes(p));
        a.ds = (char*)workspace + consts_bytes(p) + ds_table_bytes(p);
        a.ds_chunks = ds_chunks_per_head(p);
    }
    static const int ds_nt_knob = env_int("SFA_DS_NT", 1);
    a.ds_nt = ds_nt_knob;
    a.B = p.B; a.Hq = p.Hq; a.Hkv = p.Hkv; a.N = p.N;
    a.num_sink = p.num_sink;
    a.window = p.window < 0 ? 0 : (p.window > a.Nk ? a.Nk : p.window);
    a.scale = p.scale;
    a.scale_log2 = p.scale * kLog2e;
    a.q_range = slice_range(q); a.k_range = slice_range(k); a.v_range = slice_range(v); a.do_range = slice_range(d_o);
    a.dq_range = slice_range(dq); a.dk_range = slice_range(dk); a.dv_range = slice_range(dv);
    a.n_kblocks = (int)cdiv64(a.Nk, kKB);
    a.hpw = gcd(g, 8);
    a.rb = 8 / a.hpw;
    a.n_qtiles = (int)cdiv64(p.N, 32 * a.rb);
    a.hgroups = g / a.hpw;
    static const int ws_prio_knob = env_int("SFA_WS_PRIO", 1);
    a.prio = ws_prio_knob;
    if ((int64_t)a.n_qtiles * a.hgroups * p.Hkv * p.B >= (1ll << 31)) {
        set_error("bwd_mfma: grid too large");
        return SFA_ERR_UNSUPPORTED;
    }
    if (q->dtype == SFA_DTYPE_BF16) return launch_bwd_d<bf16_t>(a, p.D, p.B, stream);
    return launch_bwd_d<f16_t>(a, p.D, p.B, stream);
}

}  // namespace sfa

#ifdef SFA_STAMPS
extern "C" int sfa_debug_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sfa::g_stamps), sizeof(unsigned long long) * 64);
}
#endif
