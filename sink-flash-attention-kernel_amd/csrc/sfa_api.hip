// extern "C" entry points of libsfa.so (declared in include/sfa.h): argument
// validation, kernel-family choice, workspace carving.  No allocation, no sync.
#include <cmath>
#include <cstring>

#include "sfa_common.hpp"
#include "sfa_internal.hpp"

namespace sfa {

static thread_local char g_err[512] = "";
static char g_path[128] = "";  // diagnostic only: last kernel family dispatched in this process (any thread)

// measurement hook (sfa_debug_set_stage_events): process-global on purpose, the autograd thread runs sfa_bwd
void* g_debug_ptr = nullptr;
int g_variant[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // sfa_debug_set_variant: read by A/B builds (-DSFA_AB) only
static void* const* g_stage_events = nullptr;
static int g_stage_count = 0;
bool stage_events_armed() { return g_stage_events != nullptr; }
void record_stage(int i, hipStream_t stream) {
    if (g_stage_events && i < g_stage_count && g_stage_events[i]) (void)hipEventRecord((hipEvent_t)g_stage_events[i], stream);
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void set_path(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_path, sizeof(g_path), fmt, ap);
    va_end(ap);
}

namespace {

int check_tensor(const sfa_tensor* t, const char* name) {
    SFA_CHECK_ARG(t != nullptr, "%s: null tensor descriptor", name);
    SFA_CHECK_ARG(t->dtype == SFA_DTYPE_F32 || t->dtype == SFA_DTYPE_F16 || t->dtype == SFA_DTYPE_BF16,
                  "%s: unknown dtype %d", name, t->dtype);
    for (int i = 0; i < 4; ++i) SFA_CHECK_ARG(t->shape[i] >= 0, "%s: negative shape[%d]", name, i);
    const bool empty = t->shape[0] == 0 || t->shape[1] == 0 || t->shape[2] == 0 || t->shape[3] == 0;
    SFA_CHECK_ARG(empty || t->ptr != nullptr, "%s: null data pointer", name);
    SFA_CHECK_ARG(t->shape[3] <= 1 || t->stride[3] == 1, "%s: last-dim stride must be 1 (got %lld)", name,
                  (long long)t->stride[3]);
    for (int i = 0; i < 3; ++i)
        SFA_CHECK_ARG(t->stride[i] >= 0, "%s: negative stride[%d] not supported", name, i);
    return SFA_OK;
}

int same_shape(const sfa_tensor* a, const sfa_tensor* b, const char* an, const char* bn) {
    for (int i = 0; i < 4; ++i)
        SFA_CHECK_ARG(a->shape[i] == b->shape[i], "%s and %s differ in shape[%d]: %lld vs %lld", an, bn, i,
                      (long long)a->shape[i], (long long)b->shape[i]);
    SFA_CHECK_ARG(a->dtype == b->dtype, "%s and %s differ in dtype", an, bn);
    return SFA_OK;
}

// shape contract of SinkFlashAttentionFunc.forward (sink_flash_attention.py:494-498)
int check_prefill(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, Problem* p, int num_sink,
                  int window, float scale) {
    int st;
    if ((st = check_tensor(q, "q")) || (st = check_tensor(k, "k")) || (st = check_tensor(v, "v"))) return st;
    if ((st = same_shape(k, v, "k", "v"))) return st;
    SFA_CHECK_ARG(q->dtype == k->dtype, "q and k differ in dtype");
    SFA_CHECK_ARG(q->shape[0] == k->shape[0], "batch mismatch: q %lld vs k %lld", (long long)q->shape[0],
                  (long long)k->shape[0]);
    // the reference needs N_q == N_kv; here the keys may outnumber the queries (the queries are then the LAST N_q
    // positions: chunked prefill, a sequence-parallel rank with its halo keys prepended)
    SFA_CHECK_ARG(q->shape[2] <= k->shape[2], "prefill needs N_q <= N_kv (q %lld, k %lld)", (long long)q->shape[2],
                  (long long)k->shape[2]);
    SFA_CHECK_ARG(q->shape[3] == k->shape[3], "head dim mismatch");
    SFA_CHECK_ARG(k->shape[1] > 0 && q->shape[1] % k->shape[1] == 0, "H_q (%lld) must be divisible by H_kv (%lld)",
                  (long long)q->shape[1], (long long)k->shape[1]);
    SFA_CHECK_ARG(q->shape[2] < (1ll << 30) && q->shape[0] * q->shape[1] < (1ll << 30), "problem too large");
    SFA_CHECK_ARG(num_sink >= 0, "num_sink must be >= 0");
    SFA_CHECK_ARG(std::isfinite(scale), "scale must be finite");
    p->B = (int)q->shape[0];
    p->Hq = (int)q->shape[1];
    p->Hkv = (int)k->shape[1];
    p->N = (int)q->shape[2];
    p->Nk = (int)k->shape[2];
    p->D = (int)q->shape[3];
    p->num_sink = num_sink;
    p->window = window;
    p->scale = scale;
    return SFA_OK;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct BwdWorkspace {
    size_t delta_off, dsaux_off, mfma_off, total;
};

BwdWorkspace bwd_layout(const Problem& p, int dtype, bool use_mfma, unsigned flags) {
    BwdWorkspace w;
    size_t off = 0;
    w.delta_off = off;
    off += align256((size_t)p.B * p.Hq * p.N * sizeof(float));
    w.dsaux_off = off;
    off += align256((size_t)p.B * p.Hq * (size_t)bwd_preprocess_nblk(p.N) * sizeof(float));
    w.mfma_off = off;
    if (use_mfma) off += align256(bwd_mfma_workspace_bytes(p, dtype, flags));
    w.total = off;
    return w;
}

}  // namespace
}  // namespace sfa

using namespace sfa;

extern "C" {

int sfa_abi_version(void) { return SFA_ABI_VERSION; }
const char* sfa_last_error(void) { return g_err; }
const char* sfa_last_path(void) { return g_path; }
int sfa_debug_set_variant(int which, int value) {
    if (which < 0 || which >= 8) return SFA_ERR_INVALID_ARGUMENT;
    __atomic_store_n(&sfa::g_variant[which], value, __ATOMIC_RELAXED);
    return SFA_OK;
}

int sfa_debug_set_ptr(void* p) {
    sfa::g_debug_ptr = p;
    return SFA_OK;
}

int sfa_debug_set_stage_events(void* const* events, int count) {
    g_stage_events = count > 0 ? events : nullptr;
    g_stage_count = count > 0 ? count : 0;
    return SFA_OK;
}

int sfa_fwd(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o, float* lse,
            const float* s_aux, int num_sink, int window, float scale, unsigned flags, void* stream) {
    g_err[0] = 0;
    Problem p;
    int st = check_prefill(q, k, v, &p, num_sink, window, scale);
    if (st) return st;
    if ((st = check_tensor(o, "o")) || (st = same_shape(q, o, "q", "o"))) return st;
    if (p.B == 0 || p.Hq == 0 || p.N == 0) return SFA_OK;
    SFA_CHECK_ARG(p.D > 0, "head dim must be > 0");
    SFA_CHECK_ARG(lse != nullptr, "lse: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (!(flags & SFA_FLAG_FORCE_GENERIC) && fwd_mfma_supported(q->dtype, p.D))
        return fwd_mfma(q, k, v, o, lse, s_aux, p, s);
    if (p.Nk != p.N) {
        set_error("N_q != N_kv is served by the MFMA kernels only (16-bit dtypes, head dims 64/80/96/128)");
        return SFA_ERR_UNSUPPORTED;
    }
    return fwd_generic(q, k, v, o, lse, s_aux, p, s);
}

size_t sfa_bwd_workspace_bytes(int64_t B, int64_t Hq, int64_t Hkv, int64_t N, int64_t D, int dtype, int num_sink,
                               int window, unsigned flags) {
    Problem p{(int)B, (int)Hq, (int)Hkv, (int)N, (int)D, num_sink, window, 1.f};
    const bool use_mfma = !(flags & SFA_FLAG_FORCE_GENERIC) && bwd_mfma_supported(dtype, (int)D);
    return bwd_layout(p, dtype, use_mfma, flags).total;
}

int sfa_bwd(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o,
            const sfa_tensor* d_o, const float* lse, const float* s_aux, const sfa_tensor* dq, const sfa_tensor* dk,
            const sfa_tensor* dv, float* ds_aux, void* workspace, size_t workspace_bytes, int num_sink, int window,
            float scale, unsigned flags, void* stream) {
    g_err[0] = 0;
    Problem p;
    int st = check_prefill(q, k, v, &p, num_sink, window, scale);
    if (st) return st;
    if ((st = check_tensor(o, "o")) || (st = same_shape(q, o, "q", "o"))) return st;
    if ((st = check_tensor(d_o, "do")) || (st = same_shape(q, d_o, "q", "do"))) return st;
    if ((st = check_tensor(dq, "dq")) || (st = same_shape(q, dq, "q", "dq"))) return st;
    if ((st = check_tensor(dk, "dk")) || (st = same_shape(k, dk, "k", "dk"))) return st;
    if ((st = check_tensor(dv, "dv")) || (st = same_shape(v, dv, "v", "dv"))) return st;
    if (p.B == 0 || p.Hq == 0 || p.N == 0) return SFA_OK;
    SFA_CHECK_ARG(p.D > 0, "head dim must be > 0");
    SFA_CHECK_ARG(lse != nullptr, "lse: null pointer");
    SFA_CHECK_ARG((s_aux == nullptr) == (ds_aux == nullptr), "ds_aux must be given iff s_aux is");
    const bool use_mfma = !(flags & SFA_FLAG_FORCE_GENERIC) && bwd_mfma_supported(q->dtype, p.D);
    const BwdWorkspace w = bwd_layout(p, q->dtype, use_mfma, flags);
    if (workspace == nullptr || workspace_bytes < w.total || ((uintptr_t)workspace & 255) != 0) {
        set_error("bwd workspace: need %zu bytes, 256-byte aligned (got %zu at %p)", w.total, workspace_bytes,
                  workspace);
        return SFA_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* delta = reinterpret_cast<float*>((char*)workspace + w.delta_off);
    float* dsaux_part = reinterpret_cast<float*>((char*)workspace + w.dsaux_off);
    record_stage(0, s);
    // the row constants of the dK/dV kernel come out of the same pass whenever it is the vectorised one
    const bool fuse = use_mfma && bwd_mfma_wants_consts() && bwd_preprocess_vectorised(o, d_o, p);
    st = bwd_preprocess(o, d_o, lse, s_aux, delta, dsaux_part, ds_aux, p, s,
                        fuse ? reinterpret_cast<float*>((char*)workspace + w.mfma_off) : nullptr, bwd_mfma_lse_factor(p, flags));
    record_stage(1, s);
    if (st) return st;
    if (use_mfma)
        st = bwd_mfma(q, k, v, d_o, lse, delta, dq, dk, dv, (char*)workspace + w.mfma_off, p, flags, s, fuse);
    else if (p.Nk != p.N) {
        set_error("N_q != N_kv is served by the MFMA kernels only (16-bit dtypes, head dims 64/80/96/128)");
        st = SFA_ERR_UNSUPPORTED;
    } else
        st = bwd_generic(q, k, v, d_o, lse, delta, dq, dk, dv, p, s);
    record_stage(3, s);
    return st;
}

int sfa_varlen_supported(int dtype, int64_t D) {
    return fwd_mfma_supported(dtype, (int)D) && bwd_mfma_varlen_supported(dtype, (int)D) ? 1 : 0;
}

// shared checks of the packed entry points; on success *p describes the packed tensors (B = 1, N = total rows) and
// *run the launch (B = n_seq, N = max_seqlen, cu set)
static int check_varlen(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const int32_t* cu, int n_seq,
                        int max_seqlen, int num_sink, int window, float scale, Problem* p, Problem* run) {
    int st = check_prefill(q, k, v, p, num_sink, window, scale);
    if (st) return st;
    SFA_CHECK_ARG(p->B == 1, "packed layout: batch dim must be 1 (got %d)", p->B);
    SFA_CHECK_ARG(p->N == p->Nk, "packed layout: q and k must have the same number of rows");
    SFA_CHECK_ARG(cu != nullptr && n_seq >= 1, "cu_seqlens: need a device array of n_seq + 1 >= 2 offsets");
    SFA_CHECK_ARG(max_seqlen >= 1 && max_seqlen <= p->N, "max_seqlen %d out of range (total rows %d)", max_seqlen, p->N);
    if (!sfa_varlen_supported(q->dtype, p->D)) {
        set_error("packed (varlen) kernels exist for 16-bit dtypes and head dims 64/80/96/128 only");
        return SFA_ERR_UNSUPPORTED;
    }
    *run = *p;
    run->B = n_seq;
    run->N = max_seqlen;
    run->cu = cu;
    run->n_total = p->N;
    run->Nk = 0;
    return SFA_OK;
}

int sfa_fwd_varlen(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o, float* lse,
                   const float* s_aux, const int32_t* cu_seqlens, int n_seq, int max_seqlen, int num_sink, int window,
                   float scale, unsigned flags, void* stream) {
    g_err[0] = 0;
    (void)flags;
    Problem p, run;
    int st = check_varlen(q, k, v, cu_seqlens, n_seq, max_seqlen, num_sink, window, scale, &p, &run);
    if (st) return st;
    if ((st = check_tensor(o, "o")) || (st = same_shape(q, o, "q", "o"))) return st;
    if (p.Hq == 0 || p.N == 0) return SFA_OK;
    SFA_CHECK_ARG(lse != nullptr, "lse: null pointer");
    return fwd_mfma(q, k, v, o, lse, s_aux, run, (hipStream_t)stream);
}

int sfa_bwd_varlen(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o,
                   const sfa_tensor* d_o, const float* lse, const float* s_aux, const sfa_tensor* dq,
                   const sfa_tensor* dk, const sfa_tensor* dv, float* ds_aux, const int32_t* cu_seqlens, int n_seq,
                   int max_seqlen, void* workspace, size_t workspace_bytes, int num_sink, int window, float scale,
                   unsigned flags, void* stream) {
    g_err[0] = 0;
    (void)flags;
    Problem p, run;
    int st = check_varlen(q, k, v, cu_seqlens, n_seq, max_seqlen, num_sink, window, scale, &p, &run);
    if (st) return st;
    if ((st = check_tensor(o, "o")) || (st = same_shape(q, o, "q", "o"))) return st;
    if ((st = check_tensor(d_o, "do")) || (st = same_shape(q, d_o, "q", "do"))) return st;
    if ((st = check_tensor(dq, "dq")) || (st = same_shape(q, dq, "q", "dq"))) return st;
    if ((st = check_tensor(dk, "dk")) || (st = same_shape(k, dk, "k", "dk"))) return st;
    if ((st = check_tensor(dv, "dv")) || (st = same_shape(v, dv, "v", "dv"))) return st;
    if (p.Hq == 0 || p.N == 0) return SFA_OK;
    SFA_CHECK_ARG(lse != nullptr, "lse: null pointer");
    SFA_CHECK_ARG((s_aux == nullptr) == (ds_aux == nullptr), "ds_aux must be given iff s_aux is");
    const BwdWorkspace w = bwd_layout(p, q->dtype, true, 0);      // same carving as sfa_bwd with B = 1, N = total rows
    if (workspace == nullptr || workspace_bytes < w.total || ((uintptr_t)workspace & 255) != 0) {
        set_error("bwd workspace: need %zu bytes, 256-byte aligned (got %zu at %p)", w.total, workspace_bytes,
                  workspace);
        return SFA_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float* delta = reinterpret_cast<float*>((char*)workspace + w.delta_off);
    float* dsaux_part = reinterpret_cast<float*>((char*)workspace + w.dsaux_off);
    record_stage(0, s);
    const bool fuse = bwd_mfma_wants_consts() && bwd_preprocess_vectorised(o, d_o, p);
    st = bwd_preprocess(o, d_o, lse, s_aux, delta, dsaux_part, ds_aux, p, s,           // row-wise: no sequence structure
                        fuse ? reinterpret_cast<float*>((char*)workspace + w.mfma_off) : nullptr,
                        bwd_mfma_lse_factor(run, flags));   // (the LAUNCH problem: the kernel choice clamps the window to max_seqlen)
    record_stage(1, s);
    if (st) return st;
    st = bwd_mfma(q, k, v, d_o, lse, delta, dq, dk, dv, (char*)workspace + w.mfma_off, run, flags, s, fuse);
    record_stage(3, s);
    return st;
}

size_t sfa_decode_workspace_bytes(int64_t B, int64_t Hq, int64_t Hkv, int64_t Nkv, int64_t D, int dtype) {
    DecodePlan pl;
    if (decode_plan(B, Hq, Hkv, Nkv, D, dtype, &pl) != SFA_OK) return 0;
    // int32 counters [B * Hkv + 1] of the one-pass mode (fixed place: the start), then the split partials.  Sized for
    // pl.max_splits, the largest split count ANY key count <= Nkv can plan (the split count itself is not monotonic
    // in Nkv once the workgroup target caps it), so a workspace sized for a ring's capacity serves every fill level.
    return decode_counter_bytes(B, Hkv) + align256((size_t)B * Hq * pl.max_splits * (size_t)(D + 2) * sizeof(float));
}

}  // extern "C"

namespace {

// shared by sfa_decode (one contiguous key segment) and sfa_decode_ring (sink buffer + window ring)
int decode_common(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, int64_t n1, const sfa_tensor* k2,
                  const sfa_tensor* v2, int64_t n2, const sfa_tensor* o, const float* s_aux, void* workspace,
                  size_t workspace_bytes, float scale, void* stream, const sfa_tensor* k_new = nullptr,
                  const sfa_tensor* v_new = nullptr, int64_t new_slot = -1, int* dyn_state = nullptr,
                  unsigned flags = 0) {
    g_err[0] = 0;
    int st;
    if ((st = check_tensor(q, "q")) || (st = check_tensor(k, "k")) || (st = check_tensor(v, "v")) ||
        (st = check_tensor(o, "o")))
        return st;
    if ((st = same_shape(k, v, "k", "v")) || (st = same_shape(q, o, "q", "o"))) return st;
    SFA_CHECK_ARG(q->dtype == k->dtype, "q and k differ in dtype");
    // decode_kernel.py:146-147
    SFA_CHECK_ARG(q->shape[2] == 1, "sink_decode_attention requires N_q=1, got %lld", (long long)q->shape[2]);
    SFA_CHECK_ARG(q->shape[0] == k->shape[0] && q->shape[3] == k->shape[3], "q/k batch or head-dim mismatch");
    SFA_CHECK_ARG(k->shape[1] > 0 && q->shape[1] % k->shape[1] == 0, "H_q (%lld) must be divisible by H_kv (%lld)",
                  (long long)q->shape[1], (long long)k->shape[1]);
    SFA_CHECK_ARG(std::isfinite(scale), "scale must be finite");
    SFA_CHECK_ARG(n1 >= 0 && n1 <= k->shape[2], "first key segment: %lld valid rows of %lld", (long long)n1,
                  (long long)k->shape[2]);
    if (k2 || v2 || n2) {
        SFA_CHECK_ARG(k2 && v2, "second key segment needs both k and v");
        if ((st = check_tensor(k2, "k2")) || (st = check_tensor(v2, "v2")) || (st = same_shape(k2, v2, "k2", "v2")))
            return st;
        SFA_CHECK_ARG(k2->dtype == k->dtype && k2->shape[0] == k->shape[0] && k2->shape[1] == k->shape[1] &&
                          k2->shape[3] == k->shape[3],
                      "the two key segments must agree in dtype, batch, heads and head dim");
        SFA_CHECK_ARG(n2 >= 0 && n2 <= k2->shape[2], "second key segment: %lld valid rows of %lld", (long long)n2,
                      (long long)k2->shape[2]);
    }
    SFA_CHECK_ARG(n1 + n2 < (1ll << 31) - 4096, "N_kv too large");
    if (k_new || v_new) {       // fused cache step: the new token's K/V, stored into ring slot new_slot by the kernel
        SFA_CHECK_ARG(k_new && v_new && k2 && v2, "fused step needs k_new, v_new and the window ring");
        if ((st = check_tensor(k_new, "k_new")) || (st = check_tensor(v_new, "v_new")) ||
            (st = same_shape(k_new, v_new, "k_new", "v_new")))
            return st;
        SFA_CHECK_ARG(k_new->dtype == k2->dtype && k_new->shape[0] == k2->shape[0] && k_new->shape[1] == k2->shape[1] &&
                          k_new->shape[2] == 1 && k_new->shape[3] == k2->shape[3],
                      "k_new / v_new must be [B, H_kv, 1, D] in the ring's dtype");
        SFA_CHECK_ARG(dyn_state || (new_slot >= 0 && new_slot < n2), "write slot %lld outside the %lld valid ring slots",
                      (long long)new_slot, (long long)n2);
    }
    if (q->shape[0] == 0 || q->shape[1] == 0) return SFA_OK;
    DecodePlan pl;
    st = decode_plan(q->shape[0], q->shape[1], k->shape[1], n1 + n2, q->shape[3], q->dtype, &pl);
    if (st) return st;
    const int es = dtype_size(q->dtype);
    const sfa_tensor* ts[7] = {q, k, v, k2, v2, k_new, v_new};
    for (const sfa_tensor* t : ts) {
        if (!t || t->shape[2] == 0) continue;
        SFA_CHECK_ARG(((uintptr_t)t->ptr % 16) == 0 && (t->stride[0] * es) % 16 == 0 && (t->stride[1] * es) % 16 == 0 &&
                          (t->stride[2] * es) % 16 == 0,
                      "decode: q/k/v rows must be 16-byte aligned");
    }
    const size_t need = sfa_decode_workspace_bytes(q->shape[0], q->shape[1], k->shape[1], n1 + n2, q->shape[3], q->dtype);
    if (workspace == nullptr || workspace_bytes < need || ((uintptr_t)workspace & 255) != 0) {
        set_error("decode workspace: need %zu bytes, 256-byte aligned (got %zu at %p)", need, workspace_bytes,
                  workspace);
        return SFA_ERR_WORKSPACE;
    }
    return decode_launch(q, k, v, n1, n2 ? k2 : nullptr, n2 ? v2 : nullptr, n2, o, s_aux, workspace, scale, pl,
                         (hipStream_t)stream, k_new, v_new, (int)new_slot, dyn_state,
                         (flags & SFA_FLAG_DECODE_ONE_PASS) != 0);
}

}  // namespace

extern "C" {

int sfa_decode(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o,
               const float* s_aux, void* workspace, size_t workspace_bytes, float scale, unsigned flags,
               void* stream) {
    return decode_common(q, k, v, k ? k->shape[2] : 0, nullptr, nullptr, 0, o, s_aux, workspace, workspace_bytes, scale,
                         stream, nullptr, nullptr, -1, nullptr, flags);
}

int sfa_decode_ring(const sfa_tensor* q, const sfa_tensor* sink_k, const sfa_tensor* sink_v, int64_t sink_len,
                    const sfa_tensor* window_k, const sfa_tensor* window_v, int64_t window_len, const sfa_tensor* o,
                    const float* s_aux, void* workspace, size_t workspace_bytes, float scale, unsigned flags,
                    void* stream) {
    return decode_common(q, sink_k, sink_v, sink_len, window_k, window_v, window_len, o, s_aux, workspace,
                         workspace_bytes, scale, stream, nullptr, nullptr, -1, nullptr, flags);
}

int sfa_decode_ring_step(const sfa_tensor* q, const sfa_tensor* sink_k, const sfa_tensor* sink_v, int64_t sink_len,
                         const sfa_tensor* window_k, const sfa_tensor* window_v, int64_t window_len,
                         int64_t write_pos, const sfa_tensor* k_new, const sfa_tensor* v_new, const sfa_tensor* o,
                         const float* s_aux, void* workspace, size_t workspace_bytes, float scale, unsigned flags,
                         void* stream) {
    g_err[0] = 0;
    SFA_CHECK_ARG(k_new != nullptr && v_new != nullptr, "k_new / v_new: null tensor descriptor");
    return decode_common(q, sink_k, sink_v, sink_len, window_k, window_v, window_len, o, s_aux, workspace,
                         workspace_bytes, scale, stream, k_new, v_new, write_pos, nullptr, flags);
}

int sfa_decode_ring_step_dyn(const sfa_tensor* q, const sfa_tensor* sink_k, const sfa_tensor* sink_v,
                             const sfa_tensor* window_k, const sfa_tensor* window_v, const sfa_tensor* k_new,
                             const sfa_tensor* v_new, const sfa_tensor* o, const float* s_aux, int32_t* state,
                             void* workspace, size_t workspace_bytes, float scale, unsigned flags, void* stream) {
    g_err[0] = 0;
    SFA_CHECK_ARG(k_new != nullptr && v_new != nullptr, "k_new / v_new: null tensor descriptor");
    SFA_CHECK_ARG(state != nullptr, "state: null device pointer");
    SFA_CHECK_ARG(sink_k != nullptr && window_k != nullptr, "cache buffers: null tensor descriptor");
    // launch geometry and workspace are sized for the FULL cache (every sink row, every ring slot)
    return decode_common(q, sink_k, sink_v, sink_k->shape[2], window_k, window_v, window_k->shape[2], o, s_aux,
                         workspace, workspace_bytes, scale, stream, k_new, v_new, 0, state, flags);
}

}  // extern "C"
