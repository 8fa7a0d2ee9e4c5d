// Internal launcher declarations shared between the translation units of libsfa.
#pragma once
#include "sfa_common.hpp"

namespace sfa {

// sfa_generic.hip
int fwd_generic(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o, float* lse,
                const float* s_aux, const Problem& p, hipStream_t stream);
// consts (optional, 16-bit tensors with 16-byte aligned rows only): [B, Hq, 2, N] row constants of the wave-specialised
// dK/dV kernel, written in the same pass
int bwd_preprocess(const sfa_tensor* o, const sfa_tensor* d_o, const float* lse, const float* s_aux, float* delta,
                   float* dsaux_part, float* ds_aux, const Problem& p, hipStream_t stream, float* consts = nullptr,
                   float lse_factor = 0.f);
bool bwd_preprocess_vectorised(const sfa_tensor* o, const sfa_tensor* d_o, const Problem& p);   // can it emit consts?
int64_t bwd_preprocess_nblk(int64_t N);
int bwd_generic(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* d_o,
                const float* lse, const float* delta, const sfa_tensor* dq, const sfa_tensor* dk,
                const sfa_tensor* dv, const Problem& p, hipStream_t stream);

// sfa_fwd_mfma.hip
bool fwd_mfma_supported(int dtype, int D);
int fwd_mfma(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o, float* lse,
             const float* s_aux, const Problem& p, hipStream_t stream);

// sfa_bwd_mfma.hip
bool bwd_mfma_supported(int dtype, int D);
bool bwd_mfma_varlen_supported(int dtype, int D);   // packed (cu_seqlens) launches
size_t bwd_mfma_workspace_bytes(const Problem& p, int dtype, unsigned flags);
int bwd_mfma(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* d_o,
             const float* lse, const float* delta, const sfa_tensor* dq, const sfa_tensor* dk,
             const sfa_tensor* dv, void* workspace, const Problem& p, unsigned flags, hipStream_t stream,
             bool consts_ready = false);
bool bwd_mfma_wants_consts();   // the default dK/dV kernels read the row constants from the head of the workspace
float bwd_mfma_lse_factor(const Problem& p, unsigned flags);   // ... whose first row is -LSE * this factor (follows the dK/dV kernel choice)

// sfa_decode.hip
struct DecodePlan {
    int splits;          // KV splits per (batch, kv head)
    int keys_per_split;  // multiple of the keys one workgroup iteration covers
    int lpk;             // lanes per key row (power of two)
    int gt;              // q heads of a GQA group processed per pass
    int max_splits;      // upper bound of `splits` over every key count <= the planned one (workspace sizing)
};
// int32 arrival counters of the one-pass decode, at the start of the decode workspace
inline size_t decode_counter_bytes(int64_t B, int64_t Hkv) { return (((size_t)(B * Hkv + 1) * sizeof(int)) + 255) & ~(size_t)255; }
int decode_plan(int64_t B, int64_t Hq, int64_t Hkv, int64_t Nkv, int64_t D, int dtype, DecodePlan* plan);
// keys = rows [0, n1) of (k, v) followed by rows [0, n2) of (k2, v2); k2/v2 may be null when n2 == 0
int decode_launch(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, int64_t n1, const sfa_tensor* k2,
                  const sfa_tensor* v2, int64_t n2, const sfa_tensor* o, const float* s_aux, void* workspace,
                  float scale, const DecodePlan& plan, hipStream_t stream, const sfa_tensor* k_new = nullptr,
                  const sfa_tensor* v_new = nullptr, int new_slot = -1, int* dyn_state = nullptr,
                  bool one_pass = false);

}  // namespace sfa
