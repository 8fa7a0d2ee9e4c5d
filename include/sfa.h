/*
 * sfa.h -- C ABI of libsfa.so: MI355X (gfx950) sink flash attention.
 *
 * This is the drop-in boundary for the hot path of
 * RulinShao/sink-flash-attention-kernel.  Each entry point replaces one Triton
 * launch site (plus the eager PyTorch epilogue next to it) of the reference;
 * citations are <file>:<line> relative to the reference repository.
 *
 *   sfa_fwd      replaces  sink_attention/sink_flash_attention.py:537-554
 *                          (_sink_flash_attn_fwd_kernel, :93-194)
 *   sfa_bwd      replaces  sink_attention/sink_flash_attention.py:581-665
 *                          (Delta at :582, _sink_flash_attn_bwd_dkdv_kernel :256-364,
 *                           _sink_flash_attn_bwd_dq_kernel :371-484, the GQA group sum
 *                           :648-651 and ds_aux :658-665)
 *   sfa_decode   replaces  sink_attention/decode_kernel.py:175-226
 *                          (_decode_split_kv_kernel :28-113 and the PyTorch phase-2
 *                           reduction :201-226)
 *
 * Conventions
 *   - Plain C: pointers, sizes, ints.  No torch / C++ types cross the boundary.
 *   - The library never allocates or frees device memory and never synchronises.  The caller owns every
 *     buffer (outputs and workspaces included) and passes DEVICE pointers.
 *   - Every launch goes to the hipStream_t passed as `stream` (void* here so the
 *     header needs no HIP include); the caller makes the right device current.  One documented exception, opt-in per
 *     call: with SFA_FLAG_BWD_OVERLAP sfa_bwd may put its dQ kernel on a library-owned side stream (see the flag).
 *   - Re-entrant.  Mutable state inside the library, all of it listed here: the thread-local error / path strings; a
 *     per-kernel bitmask of the devices whose dynamic-LDS attribute has been set (atomic OR, idempotent); and, only
 *     for callers that pass SFA_FLAG_BWD_OVERLAP, one side stream + two events per (calling thread, device), created
 *     at that thread's first overlapped sfa_bwd on the device and destroyed when the thread ends.  Nothing else
 *     survives a call.  A RELEASE build reads no environment variable and ignores sfa_debug_set_variant: the SFA_*
 *     tuning knobs and the A/B variants exist in -DSFA_AB development builds only (tools/ab.py, tools/build_ab.sh).
 *   - Return value: 0 = ok, <0 = SFA_ERR_* (argument / support problem, nothing
 *     was launched), >0 = a hipError_t from a launch.
 *   - Tensors are described by sfa_tensor: 4-D [B, H, N, D] with strides in
 *     ELEMENTS.  Any B/H/N strides are accepted (so a [B, N, H, D] activation can be
 *     passed as a permuted view without a copy); the D stride must be 1.
 */
#ifndef SFA_H
#define SFA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFA_ABI_VERSION 2

/* element types of q/k/v/o/do/dq/dk/dv (all tensors of one call share one dtype) */
#define SFA_DTYPE_F32 0
#define SFA_DTYPE_F16 1
#define SFA_DTYPE_BF16 2

#define SFA_OK 0
#define SFA_ERR_INVALID_ARGUMENT (-1)  /* null pointer, shape mismatch, bad stride ...   */
#define SFA_ERR_UNSUPPORTED (-2)       /* e.g. head dim beyond what any kernel handles   */
#define SFA_ERR_WORKSPACE (-3)         /* workspace too small / null                      */

/* flags */
#define SFA_FLAG_FORCE_GENERIC 0x1u /* use the exact-f32 generic kernels even when an MFMA kernel exists */
/* 0x2u was SFA_FLAG_BWD_SPILL_DS in ABI revisions before round 2 (dS saved by the dK/dV kernel, dQ as a GEMM over it:
 * break-even against 7 GB of workspace, removed).  The bit is accepted and ignored. */
/* sfa_decode*: one launch instead of two.  The last KV split of a (batch, KV head) to finish folds the split partials
 * itself (atomic arrival counters).  Contract: the caller OWNS the workspace across calls and zero-initialised its FIRST
 * align256((B * Hkv + 1) * 4) bytes once; the kernel leaves them zero.  No agent-scope fence is involved (the partials are
 * written through and fetched past the per-XCD L2s, the counters are relaxed atomics).  Measured: a gain only where the cache
 * is short enough for ONE split (B = 1, 132 keys: 15.8 vs 19.2 us); at 4100 keys two launches are 2 us faster. */
#define SFA_FLAG_DECODE_ONE_PASS 0x4u
/* sfa_bwd / sfa_bwd_varlen: the dQ and dK/dV kernels are independent; on a SMALL grid (dQ grid below 4 workgroups per
 * CU: batch 1, tensor- or sequence-parallel shards) the dQ kernel is launched on a library-owned side stream of the
 * lowest queue priority, forked from and joined back into `stream` by events, so that it fills the CUs the dK/dV
 * kernel leaves idle (+1.4 ... 17 %).  Capturable (fork / join are event edges), EXCEPT that the side stream itself
 * cannot be created under capture: a thread whose FIRST overlapped call on a device happens while `stream` is being
 * captured runs that call (and every later captured one, until an uncaptured call has created the stream) without
 * overlap.  Every return path behind the fork joins the side stream again.  Without the flag every launch goes to
 * `stream` and the library creates nothing.  sink_attention (Python) sets it by default. */
#define SFA_FLAG_BWD_OVERLAP 0x8u
/* sfa_bwd / sfa_bwd_varlen / sfa_bwd_workspace_bytes: dispatch override of the dK/dV kernel for head dims 64 ... 128
 * (pass the same bits to the workspace query): _ASM = the hand-placed 256-key-block kernel wherever its body serves the
 * shape, _WS = the wave-specialised compiled kernel (128-key blocks).  Neither: the library's rule (hand-placed for
 * dense self-attention batches and for grids that fill the chip).  The parity tests run every backward case under the
 * rule AND with the hand-placed kernel forced. */
#define SFA_FLAG_BWD_DKDV_ASM 0x10u
#define SFA_FLAG_BWD_DKDV_WS 0x20u

typedef struct sfa_tensor {
    void* ptr;         /* device pointer to element [0,0,0,0]                    */
    int64_t shape[4];  /* B, H, N, D                                             */
    int64_t stride[4]; /* in elements; stride[3] must be 1                       */
    int32_t dtype;     /* SFA_DTYPE_*                                            */
    int32_t reserved;
} sfa_tensor;

int sfa_abi_version(void);

/* Thread-local description of the last error returned on this thread ("" if none). */
const char* sfa_last_error(void);

/* Name of the kernel family the last successful call in this PROCESS dispatched to (any
 * thread: autograd runs backward on its own thread), e.g. "fwd_mfma_bf16_d128_nw8_hpw4",
 * "bwd_generic_f32math".  Diagnostic for tests/benchmarks only; racy by design. */
const char* sfa_last_path(void);

/*
 * Measurement hook (bench.py): arm `count` caller-created hipEvent_t handles (null entries are
 * skipped; count = 0 disarms).  While armed, sfa_bwd records on its stream
 *   events[0] at entry, [1] after the preprocess kernels (Delta, ds_aux),
 *   [2] after the dK/dV kernel, [3] after the dQ kernel (= end of sfa_bwd).
 * Process-global and not thread-safe by design: a diagnostic, never used by the op itself.
 */
/* Development hook: in a library built with -DSFA_AB two variants of a kernel body are compiled side by side and
 * `value` picks the one `which` (0 dK/dV, 1 forward, 2 dQ) launches; 3 = workgroup order of the dK/dV kernel, 7 = its
 * ticketed dispatch off - so that tools/ab.py can time them alternately in one process on one device.  A release
 * build stores the value and never reads it: no effect on dispatch (use the SFA_FLAG_BWD_DKDV_* bits for that). */
int sfa_debug_set_variant(int which, int value);
/* Device buffer for the cycle stamps of a diagnostic build (tools/stamps_dkdv.py, tools/stamps_wl.py); unused otherwise. */
int sfa_debug_set_ptr(void* device_buffer);
int sfa_debug_set_stage_events(void* const* events, int count);

/*
 * Forward.  valid(i,j) = (j <= i) && (j < num_sink || j >= i - window + 1)
 *   q,o  [B, Hq, N, D]     k,v [B, Hkv, N, D]   Hq % Hkv == 0
 *        k,v may hold MORE rows than q (N_kv >= N): the queries are then the LAST N positions of the key sequence,
 *        row i sits at position i + N_kv - N (chunked prefill, a sequence-parallel rank with its halo keys
 *        prepended; the reference asserts N_q == N_kv).  MFMA kernels only: SFA_ERR_UNSUPPORTED for fp32 / other
 *        head dims.  dk, dv of sfa_bwd are [B, Hkv, N_kv, D]; lse, dq follow q.
 *   lse  [B, Hq, N] float32, contiguous: log-sum-exp of the scaled scores of the row
 *        INCLUDING the s_aux logit (-inf for a row that sees nothing)
 *   s_aux  nullable, [Hq] float32: per-head extra logit that only enters the denominator
 *   scale  softmax scale (the reference hard-wires 1/sqrt(D), sink_flash_attention.py:505)
 */
int sfa_fwd(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o,
            float* lse, const float* s_aux, int num_sink, int window, float scale,
            unsigned flags, void* stream);

/*
 * Backward.  Inputs q,k,v,o,do,lse,s_aux as produced/consumed by sfa_fwd.
 *   dq [B,Hq,N,D]; dk,dv [B,Hkv,N,D] (already summed over the GQA group);
 *   ds_aux nullable [Hq] float32 (required non-null iff s_aux non-null).
 *   workspace: sfa_bwd_workspace_bytes() bytes of device scratch, 256-byte aligned (N = query rows; the size covers
 *   Delta, the ds_aux partials, the row constants and, where the dK/dV sweep is split into chunks, their partial dK / dV:
 *   the query is an upper bound for every N_kv >= N).  Needs no initialisation, holds no state between calls.
 *   Deterministic: the same inputs give bitwise-identical outputs (partial sums are added in a fixed order).
 */
size_t sfa_bwd_workspace_bytes(int64_t B, int64_t Hq, int64_t Hkv, int64_t N, int64_t D, int dtype,
                               int num_sink, int window, unsigned flags);

int sfa_bwd(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o,
            const sfa_tensor* d_o, const float* lse, const float* s_aux, const sfa_tensor* dq,
            const sfa_tensor* dk, const sfa_tensor* dv, float* ds_aux, void* workspace,
            size_t workspace_bytes, int num_sink, int window, float scale, unsigned flags,
            void* stream);

/*
 * Packed (variable-length) batches, SURVEY.md section 8 f-3.  The reference cannot do this: its boundary hands packed
 * batches back to stock flash attention, which drops s_aux (sink_attention/verl_patch.py:73-93).
 *   q,o,do,dq [1, Hq, T, D]   k,v,dk,dv [1, Hkv, T, D]   lse [Hq, T] float32
 *   cu_seqlens: DEVICE int32 [n_seq + 1], cu[0] = 0, cu[n_seq] = T; sequence i = rows cu[i] .. cu[i+1].  Every
 *   sequence gets its own mask origin (valid(i, j) in its own positions) and its own s_aux term.
 *   max_seqlen: longest sequence (only sizes the grids; an upper bound is fine).
 *   The device array is NOT validated (no host synchronisation): cu must be non-decreasing with cu[0] = 0 and
 *   max_seqlen must not understate the longest sequence, or tiles are silently dropped.  Rows behind cu[n_seq]
 *   (padding) are never written: the caller zero-fills outputs / gradients if it hands in such a pack.
 *   Workspace of sfa_bwd_varlen: sfa_bwd_workspace_bytes(1, Hq, Hkv, T, D, dtype, num_sink, window, 0).
 * 16-bit dtypes and head dims 64 / 80 / 96 / 128 only (sfa_varlen_supported); SFA_ERR_UNSUPPORTED otherwise - the
 * caller can then run the sequences one by one through sfa_fwd / sfa_bwd on strided views.
 */
int sfa_varlen_supported(int dtype, int64_t D);

int sfa_fwd_varlen(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o, float* lse,
                   const float* s_aux, const int32_t* cu_seqlens, int n_seq, int max_seqlen, int num_sink,
                   int window, float scale, unsigned flags, void* stream);

int sfa_bwd_varlen(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o,
                   const sfa_tensor* d_o, const float* lse, const float* s_aux, const sfa_tensor* dq,
                   const sfa_tensor* dk, const sfa_tensor* dv, float* ds_aux, const int32_t* cu_seqlens, int n_seq,
                   int max_seqlen, void* workspace, size_t workspace_bytes, int num_sink, int window, float scale,
                   unsigned flags, void* stream);

/*
 * Single-query decode over every key handed in (no mask; windowing is the cache's job,
 * decode_kernel.py:72-83).
 *   q,o [B, Hq, 1, D]   k,v [B, Hkv, Nkv, D]   s_aux nullable [Hq] float32
 */
/* Monotonic in Nkv: a workspace sized for Nkv serves every call with the same (B, Hq, Hkv, D, dtype) and FEWER keys
 * (a ring cache sizes it once for num_sink + window_size and steps through every fill level). */
size_t sfa_decode_workspace_bytes(int64_t B, int64_t Hq, int64_t Hkv, int64_t Nkv, int64_t D,
                                  int dtype);

int sfa_decode(const sfa_tensor* q, const sfa_tensor* k, const sfa_tensor* v, const sfa_tensor* o,
               const float* s_aux, void* workspace, size_t workspace_bytes, float scale,
               unsigned flags, void* stream);

/*
 * Decode over a sink + sliding-window KV cache WITHOUT linearising it.  Replaces the
 * get_kv() torch.cat copies of sink_attention/cache.py:185-216 followed by sink_decode_attention: the kernel reads
 * rows [0, sink_len) of the sink buffer and rows [0, window_len) of the window ring in place (softmax does not care
 * about key order, so a wrapped ring needs no reordering).
 *   sink_k/v   [B, Hkv, num_sink, D]     sink_len   <= num_sink   valid rows
 *   window_k/v [B, Hkv, window_size, D]  window_len <= window_size valid slots (all of them once the ring is full)
 *   workspace: sfa_decode_workspace_bytes(B, Hq, Hkv, sink_len + window_len, D, dtype)
 */
int sfa_decode_ring(const sfa_tensor* q, const sfa_tensor* sink_k, const sfa_tensor* sink_v, int64_t sink_len,
                    const sfa_tensor* window_k, const sfa_tensor* window_v, int64_t window_len,
                    const sfa_tensor* o, const float* s_aux, void* workspace, size_t workspace_bytes,
                    float scale, unsigned flags, void* stream);

/*
 * One generation step on the sink + ring cache in ONE pass: store the new token's K/V [B, Hkv, 1, D] into ring slot
 * `write_pos` AND attend q over the cache as it is after that store.  Replaces SinkCacheLayer.update()'s slot write +
 * torch.cat linearisation (sink_attention/cache.py:129-216) followed by sink_decode_attention.
 *   window_len: valid ring slots AFTER the append (min(old + 1, window_size)); 0 <= write_pos < window_len.
 *   The slot's previous content (the evicted token) is never read; the caller advances write_pos / window_len.
 *   workspace: sfa_decode_workspace_bytes(B, Hq, Hkv, sink_len + window_len, D, dtype)
 */
int sfa_decode_ring_step(const sfa_tensor* q, const sfa_tensor* sink_k, const sfa_tensor* sink_v, int64_t sink_len,
                         const sfa_tensor* window_k, const sfa_tensor* window_v, int64_t window_len,
                         int64_t write_pos, const sfa_tensor* k_new, const sfa_tensor* v_new, const sfa_tensor* o,
                         const float* s_aux, void* workspace, size_t workspace_bytes, float scale, unsigned flags,
                         void* stream);

/*
 * The same step with the cache state on the DEVICE, so that a whole generation step (all layers) can be captured
 * into a hipGraph and replayed without host work: `state` = int32 {sink_len, window_len, write_pos}.  The call stores
 * k_new / v_new into slot write_pos, attends over sink rows [0, sink_len) and ring slots [0, min(window_len + 1,
 * window_size)) and then advances the state (window_len saturates at window_size, write_pos wraps).  Launch geometry
 * does not depend on the state: workspace = sfa_decode_workspace_bytes(B, Hq, Hkv, num_sink + window_size, D, dtype).
 */
int sfa_decode_ring_step_dyn(const sfa_tensor* q, const sfa_tensor* sink_k, const sfa_tensor* sink_v,
                             const sfa_tensor* window_k, const sfa_tensor* window_v, const sfa_tensor* k_new,
                             const sfa_tensor* v_new, const sfa_tensor* o, const float* s_aux, int32_t* state,
                             void* workspace, size_t workspace_bytes, float scale, unsigned flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SFA_H */
