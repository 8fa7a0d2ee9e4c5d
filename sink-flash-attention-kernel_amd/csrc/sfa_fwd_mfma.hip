// MFMA forward for gfx950: sink + sliding-window flash attention with s_aux.
// Replaces _sink_flash_attn_fwd_kernel (sink_attention/sink_flash_attention.py:93-194 of the reference).
//
// Work decomposition (not the reference's one-program-per-(q-block, q-head)):
//   * a workgroup = NW wavefronts = (b, KV head, q-tile) x HPW query heads of the GQA group: wave w owns
//     32 query rows of head (w % HPW), row block (w / HPW).  Every K/V tile is staged in LDS ONCE for all
//     heads/row blocks of the workgroup (the reference re-reads K/V per q head).
//   * per 64-key tile and wave: S^T = K * Q^T with mfma_f32_32x32x16 (K rows from LDS as the A operand, the
//     wave's Q rows held in registers as the B operand), so the lane owns ONE query row: running max / sum
//     (seeded with the s_aux logit) and the rescale factor are per-lane scalars, no cross-lane reductions
//     except one half-wave swap.  P^T stays in the accumulator layout and is fed back as the B operand of
//     O^T += V^T * P^T; V^T fragments come from ds_read_b64_tr_b16 on a row-major V tile.
//   * two-range walk: sink tiles [0, ceil(ns/64)) then window tiles; a tile is classified per wave as
//     skipped / fully valid (no predicate work) / edge (per-element predicate).  The reference evaluates the
//     mask on every tile (sink_flash_attention.py:64-66).
//   * K/V tiles: buffer_load (rows >= N read as zero) -> registers -> XOR-swizzled LDS, double buffered, the
//     next tile's loads are issued before the current tile's math and written after it.
#include "sfa_common.hpp"
#include "sfa_internal.hpp"

#include <cstdlib>

namespace sfa {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

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

struct FwdArgs {
    View q, k, v, o;
    float* lse;
    const float* s_aux;
    int B, Hq, Hkv, N;
    int num_sink, window;  // window already clamped to [0, N]
    float scale_log2;      // softmax scale * log2(e)
    int hpw, rb;           // q heads per workgroup, 32-row blocks per workgroup (hpw * rb == NW)
    int n_qtiles, hgroups;
    unsigned q_range, k_range, v_range, o_range;  // byte extent of one (b, head) slice
    int prio;              // tuning knob (SFA_FWD_PRIO): raised priority for waves 4..7
    const int* cu;         // packed batches: device row offsets [B + 1] (else null), see Problem
    int n_total;           // rows of the packed tensors
    int Nk;                // key rows (>= N): query row i sits at position i + Nk - N (== N for packed batches)
};

__device__ __forceinline__ float half_swap_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <typename T, int D, int NW>
__global__ __launch_bounds__(NW * 64, 2) void fwd_mfma_kernel(FwdArgs a) {
    using M = Mma<T>;
    using frag = typename M::frag;
    constexpr int DK = D / 16;          // k-steps of the QK^T contraction
    constexpr int DVB = (D + 31) / 32;  // 32-wide output column blocks of P*V
    constexpr int CPR = D / 8;          // 16-byte chunks per row
    constexpr int ROWB = (D <= 64) ? 128 : 256;
    constexpr int TILE_BYTES = 64 * ROWB;
    constexpr int NT = NW * 64;
    constexpr int NCH = 64 * CPR;
    constexpr int NLD = (NCH + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- XCD-aware work id: blocks b and b+8 share an XCD (speed only); give each XCD a contiguous work range
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    // q tiles from the END of the sequence first: a late tile walks the full window (or, causal, everything before it),
    // an early one a few key tiles, so the short workgroups should be the ones that finish the kernel
    const int qt = a.n_qtiles - 1 - bid % a.n_qtiles;
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

    // ---- tile list: sink tiles [0, ts_hi) then window tiles [tw_lo, tw_hi)
    const int ns_eff = ns < q1 + P ? ns : q1 + P;
    const int ts_hi = (ns_eff + 63) >> 6;
    int wlo = q0 + P - W + 1;
    if (wlo < 0) wlo = 0;
    int tw_lo = wlo >> 6;
    if (tw_lo < ts_hi) tw_lo = ts_hi;
    const int tw_hi = (q1 + P + 63) >> 6;
    if (tw_lo > tw_hi) tw_lo = tw_hi;
    const int nt = ts_hi + (tw_hi - tw_lo);

    // ---- buffer descriptors (wave-uniform)
    const char* qb = a.q.ptr + ((int64_t)sq.bb * a.q.sb + (int64_t)head * a.q.sh + (int64_t)sq.row0 * a.q.sn) * 2;
    const char* kb = a.k.ptr + ((int64_t)sq.bb * a.k.sb + (int64_t)hk * a.k.sh + (int64_t)sq.row0 * a.k.sn) * 2;
    const char* vb = a.v.ptr + ((int64_t)sq.bb * a.v.sb + (int64_t)hk * a.v.sh + (int64_t)sq.row0 * a.v.sn) * 2;
    char* ob = a.o.ptr + ((int64_t)sq.bb * a.o.sb + (int64_t)head * a.o.sh + (int64_t)sq.row0 * a.o.sn) * 2;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)qb, 0, seq_range(a.cu, a.q_range, N, a.q.sn, D), 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, seq_range(a.cu, a.k_range, N, a.k.sn, D), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, seq_range(a.cu, a.v_range, N, a.v.sn, D), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)ob, 0, seq_range(a.cu, a.o_range, N, a.o.sn, D), 0x00020000);

    // ---- Q fragments: B operand of S^T = K Q^T; lane (r,h) holds Q[qrow][16ks + 8h .. +8)
    frag qf[DK];
#pragma unroll
    for (int ks = 0; ks < DK; ++ks) {
        const unsigned off = (unsigned)qrow * (unsigned)(a.q.sn * 2) + (unsigned)((2 * ks + h) * 16);
        qf[ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rq, off, 0, 0));
    }

    // ---- staging: chunk c = tid + i*NT of the tile -> (key = c / CPR, ch = c % CPR)
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
            const int ksw = (ROWB == 256) ? (key & 15) : ((key >> 1) & 7);
            const int vsw = (ROWB == 256) ? ((key & 3) << 2) : (((key >> 1) & 1) << 2);
            if ((NCH % NT == 0) || (c < NCH)) {
                *reinterpret_cast<u32x4*>(kl + key * ROWB + ((ch ^ ksw) << 4)) = kst[i];
                *reinterpret_cast<u32x4*>(vl + key * ROWB + ((ch ^ vsw) << 4)) = vst[i];
            }
        }
    };
    auto tile_of = [&](int it) { return it < ts_hi ? it : tw_lo + (it - ts_hi); };

    // ---- per-lane LDS read addressing
    const int ksw_l = (ROWB == 256) ? (r & 15) : ((r >> 1) & 7);   // swizzle of key row (32*kh + r)
    const int k_rowoff = r * ROWB;
    const int q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = (lane >> 4) & 1;
    const int vsw_l = (ROWB == 256) ? (q4 << 2) : (((q4 >> 1) & 1) << 2);
    const int v_rowoff = (4 * h + q4) * ROWB + (p4 & 1) * 8;
    const int v_chunk_lo = 2 * g1 + (p4 >> 1);

    // ---- online softmax state (log2 domain); the two half-waves of a row keep partial sums of l
    float m = -INFINITY, l = 0.f;
    if (a.s_aux) {
        m = a.s_aux[head] * kLog2e;
        l = (h == 0) ? 1.f : 0.f;
    }
    f32x16 o[DVB];
#pragma unroll
    for (int db = 0; db < DVB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    const float c = a.scale_log2;
    // waves 4..7 are dispatched second and lose VALU/MFMA arbitration to their SIMD partners by age (measured with
    // s_memtime stamps: they set the tile time while waves 0..3 idle at the barrier); optional static priority raise
    if (NW == 8 && a.prio && wave >= 4) __builtin_amdgcn_s_setprio(1);

    if (nt > 0) {
        issue_loads(tile_of(0));
        write_lds(0);
    }
    // Q fragments landed, on EVERY path into the loop: without this the waitcnt pass still counts them as pending on
    // the nt == 0 edge and the merged state makes it emit vmcnt(7..0) inside the QK^T chain, which drains the next
    // tile's prefetch (issued a few instructions earlier) in the middle of every tile (C3 -1 %, causal -2 %)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
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
            // S^T tiles: rows = 32 keys (two halves), cols = the wave's 32 query rows
            // K fragments in bursts (one 32-key half ahead of the MFMA chain that consumes it): left alone, hipcc sinks
            // every ds_read next to its MFMA and the wave pays the LDS latency once per k-step
            f32x16 s[2];
            frag kfa[DK], kfb[DK];
#pragma unroll
            for (int ks = 0; ks < DK; ++ks)
                kfa[ks] = *reinterpret_cast<const frag*>(kl + k_rowoff + (((2 * ks + h) ^ ksw_l) << 4));
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s[0][i] = 0.f;
                s[1][i] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) {
                kfb[ks] = *reinterpret_cast<const frag*>(kl + 32 * ROWB + k_rowoff + (((2 * ks + h) ^ ksw_l) << 4));
                s[0] = M::run(kfa[ks], qf[ks], s[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < DK; ++ks) s[1] = M::run(kfb[ks], qf[ks], s[1]);
            if (!full) {
                int qm = qrow + P;
                asm volatile("; edge tile" : "+v"(qm) : : "memory");   // side effect: the branch cannot be speculated / if-converted
                // Most edge tiles need only ONE of the two conditions: a diagonal tile whose keys are all inside the
                // window of all the wave's rows (causal test only), or a window-boundary tile whose keys are all
                // causal (window test only).  One compare against a per-lane threshold instead of three per element.
                const bool no_sink = k0 >= ns;
                const bool all_in_window = k0 >= pw_hi - W + 1;
                const bool all_causal = k0 + 63 <= pw0;
                if (no_sink && all_in_window) {
                    const int t = qm - k0 - 4 * h;                 // valid iff (32 kh + (i&3) + 8 (i>>2)) <= t
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                        for (int i = 0; i < 16; ++i) s[kh][i] = (32 * kh + (i & 3) + 8 * (i >> 2) <= t) ? s[kh][i] : -INFINITY;
                } else if (no_sink && all_causal) {
                    const int t = qm - W - k0 - 4 * h;             // valid iff (32 kh + ...) > t
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                        for (int i = 0; i < 16; ++i) s[kh][i] = (32 * kh + (i & 3) + 8 * (i >> 2) > t) ? s[kh][i] : -INFINITY;
                } else {
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int key = k0 + 32 * kh + (i & 3) + 8 * (i >> 2) + 4 * h;
                            const bool valid = (key <= qm) && (key < ns || key + W > qm);
                            s[kh][i] = valid ? s[kh][i] : -INFINITY;
                        }
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kh][i]);
            mx = half_swap_max(mx);
            // Deferred rescale: keep the old reference point unless the new row maximum is more than 2^8 above it (or
            // there was none yet).  p then ranges up to 2^8 instead of 1 - harmless in f32 / bf16 - and the rescale of
            // O below, which otherwise fires in about half of all tiles (some row of the wave sets a record), becomes
            // rare after the first tiles.  LSE = m + log2(l) is unchanged by the choice of m.
            const float m_cand = fmaxf(m, mx * c);
            const float m_new = (m_cand > m + 8.f || m == -INFINITY) ? m_cand : m;
            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m - m_safe);
            // packed f32 math (v_pk_fma_f32 / v_pk_add_f32: two elements per issue slot) for the exponent argument and
            // the row sum; the kernel issues ~8 VALU instructions per MFMA, so VALU issue slots matter as much as MFMAs
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 c2 = {c, c}, nm2 = {-m_safe, -m_safe};
            f32x2 rs2 = {0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x2 sv = {s[kh][i], s[kh][i + 1]};
                    const f32x2 x = __builtin_elementwise_fma(sv, c2, nm2);
                    f32x2 p;
                    p[0] = __builtin_amdgcn_exp2f(x[0]);
                    p[1] = __builtin_amdgcn_exp2f(x[1]);
                    s[kh][i] = p[0];
                    s[kh][i + 1] = p[1];
                    rs2 += p;
                }
            const float rs = rs2[0] + rs2[1];
            l = fmaf(l, alpha, rs);
            m = m_new;
            if (!__all(alpha == 1.f)) {
#pragma unroll
                for (int db = 0; db < DVB; ++db)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
            }
            // O^T += V^T P^T : P^T accumulators are the B operand (k index = key, permuted: see probes.hip #4)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    frag pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (typename M::elem)s[kh][8 * st + j];
                    const char* vrow = vl + (32 * kh + 16 * st) * ROWB + v_rowoff;
#pragma unroll
                    for (int db = 0; db < DVB; ++db) {
                        const int ch = (4 * db + v_chunk_lo) ^ vsw_l;
                        const char* p1 = vrow + (ch << 4);
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (s16x4 __attribute__((address_space(3)))*)(p1));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (s16x4 __attribute__((address_space(3)))*)(p1 + 8 * ROWB));
                        s16x8 vv;
                        vv[0] = lo[0]; vv[1] = lo[1]; vv[2] = lo[2]; vv[3] = lo[3];
                        vv[4] = hi[0]; vv[5] = hi[1]; vv[6] = hi[2]; vv[7] = hi[3];
                        o[db] = M::run(__builtin_bit_cast(frag, vv), pf, o[db]);
                    }
                }
        }
        if (it + 1 < nt) write_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: O = acc / l (l == 0 -> 1), LSE = ln2 * (m + log2 l)   (sink_flash_attention.py:183-194)
    float lt = half_swap_sum(l);
    if (lt == 0.f) lt = 1.f;
    const float inv = 1.f / lt;
    const unsigned orow = (unsigned)qrow * (unsigned)(a.o.sn * 2);
    // Widened store tail: lane (row, h) holds d = 8*g4 + 4*h + {0..3} of every 32-wide block, i.e. 8-byte pieces.  One
    // v_permlane32_swap per register hands the two half-waves each other's half of a 16-byte piece (h = 0 takes the
    // even g4, h = 1 the odd one), so the row goes out as 16-byte stores: half the store instructions.
#pragma unroll
    for (int db = 0; db < DVB; ++db)
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            if (32 * db + 16 * p2 < D) {
                typedef typename M::elem E;
                typedef __attribute__((ext_vector_type(4))) E e4;
                e4 pe, po;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pe[e] = (E)(o[db][8 * p2 + e] * inv);
                    po[e] = (E)(o[db][8 * p2 + 4 + e] * inv);
                }
                const u32x2 ev = __builtin_bit_cast(u32x2, pe), od = __builtin_bit_cast(u32x2, po);
                const auto s0 = __builtin_amdgcn_permlane32_swap(ev[0], od[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(ev[1], od[1], false, false);
                u32x4 w;
                w[0] = s0[0]; w[1] = s1[0]; w[2] = s0[1]; w[3] = s1[1];
                const int d = 32 * db + 16 * p2 + 8 * h;
                const unsigned off = wave_live ? orow + (unsigned)(d * 2) : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_buffer_store_b128(w, ro, off, 0, 0);
            }
        }
    if (h == 0 && qrow < N) {
        const int64_t lrow = a.cu ? (int64_t)head * a.n_total + sq.row0 : ((int64_t)b * a.Hq + head) * N;
        a.lse[lrow + qrow] = (m + __log2f(lt)) * kLn2;
    }
}

template <typename T, int D, int NW>
int launch(const FwdArgs& a, int nblk, hipStream_t stream) {
    constexpr int ROWB = (D <= 64) ? 128 : 256;
    constexpr int lds = 2 * 2 * 64 * ROWB;
    auto kern = fwd_mfma_kernel<T, D, NW>;
    static unsigned long long attr_done = 0;
    ensure_dynamic_lds((const void*)kern, lds, &attr_done);
    kern<<<dim3(nblk), dim3(NW * 64), lds, stream>>>(a);
    set_path("fwd_mfma_%s_d%d_nw%d_hpw%d", sizeof(T) == 2 && DT<T>::id == SFA_DTYPE_BF16 ? "bf16" : "f16", D, NW,
             a.hpw);
    return launch_status("fwd_mfma");
}

template <typename T, int NW>
int launch_d(const FwdArgs& a, int D, int nblk, hipStream_t stream) {
    switch (D) {
        case 64: return launch<T, 64, NW>(a, nblk, stream);
        case 80: return launch<T, 80, NW>(a, nblk, stream);
        case 96: return launch<T, 96, NW>(a, nblk, stream);
        case 128: return launch<T, 128, NW>(a, nblk, stream);
    }
    set_error("fwd_mfma: unsupported head dim %d", D);
    return SFA_ERR_UNSUPPORTED;
}

int gcd(int x, int y) { return y == 0 ? x : gcd(y, x % y); }

unsigned slice_range(const sfa_tensor* t) {
    return (unsigned)(((t->shape[2] - 1) * t->stride[2] + t->shape[3]) * 2);
}
bool slice_ok(const sfa_tensor* t) {
    // 32-bit buffer offsets must cover rows up to N + 63; rows 16-byte aligned
    const int64_t reach = (t->shape[2] + 64) * t->stride[2] * 2 + 512;
    return reach < (1ll << 32) - 65536 && ((uintptr_t)t->ptr % 16) == 0 && (t->stride[0] * 2) % 16 == 0 &&
           (t->stride[1] * 2) % 16 == 0 && (t->stride[2] * 2) % 16 == 0;
}

}  // namespace

bool fwd_mfma_supported(int dtype, int D) {
    return (dtype == SFA_DTYPE_BF16 || dtype == SFA_DTYPE_F16) && (D == 64 || D == 80 || D == 96 || D == 128);
}

int fwd_mfma(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o, float* lse,
             const float* s_aux, const Problem& p, hipStream_t stream) {
    if (!(slice_ok(q) && slice_ok(k) && slice_ok(v) && slice_ok(o))) {
        if (p.cu || (p.Nk > 0 && p.Nk != p.N)) {
            set_error("packed (varlen) / N_q != N_kv forward needs 16-byte aligned rows and < 4 GiB head slices");
            return SFA_ERR_UNSUPPORTED;
        }
        return fwd_generic(q, k, v, o, lse, s_aux, p, stream);  // unaligned / >4 GiB slices: exact path
    }
    // 8 waves (one workgroup per CU) win when a workgroup walks many key tiles; with a short window (few tiles per
    // workgroup, prologue / epilogue bound) two 4-wave workgroups per CU overlap each other's latencies:
    // measured C2 (W=1024) +7 %, gpt-oss sliding layers (W=128) +8..12 %, C3 (W=4096) -3 %
    int NW = (p.window >= 0 && p.window <= 2048) ? 4 : 8;
    static const int nw_knob = env_int("SFA_FWD_NW", 0);
    if (nw_knob == 4 || nw_knob == 8) NW = nw_knob;
    const int g = p.Hq / p.Hkv;
    FwdArgs a;
    a.q = make_view(q); a.k = make_view(k); a.v = make_view(v); a.o = make_view(o);
    a.lse = lse; a.s_aux = s_aux;
    a.B = p.B; a.Hq = p.Hq; a.Hkv = p.Hkv; a.N = p.N;
    a.cu = p.cu; a.n_total = p.n_total;
    a.Nk = p.cu ? p.N : (p.Nk > 0 ? p.Nk : p.N);
    a.num_sink = p.num_sink;
    a.window = p.window < 0 ? 0 : (p.window > a.Nk ? a.Nk : p.window);
    a.scale_log2 = p.scale * kLog2e;
    a.hpw = gcd(g, NW);
    a.rb = NW / a.hpw;
    a.n_qtiles = (int)cdiv64(p.N, 32 * a.rb);
    a.hgroups = g / a.hpw;
    static const int prio_knob = env_int("SFA_FWD_PRIO", 1);
    a.prio = prio_knob;
    a.q_range = slice_range(q); a.k_range = slice_range(k); a.v_range = slice_range(v); a.o_range = slice_range(o);
    const int64_t nblk = (int64_t)a.n_qtiles * a.hgroups * p.Hkv * p.B;
    if (nblk >= (1ll << 31)) {
        set_error("fwd_mfma: grid too large");
        return SFA_ERR_UNSUPPORTED;
    }
    if (q->dtype == SFA_DTYPE_BF16)
        return NW == 8 ? launch_d<bf16_t, 8>(a, p.D, (int)nblk, stream) : launch_d<bf16_t, 4>(a, p.D, (int)nblk, stream);
    return NW == 8 ? launch_d<f16_t, 8>(a, p.D, (int)nblk, stream) : launch_d<f16_t, 4>(a, p.D, (int)nblk, stream);
}

}  // namespace sfa
