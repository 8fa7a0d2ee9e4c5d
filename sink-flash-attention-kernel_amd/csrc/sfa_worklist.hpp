// Work lists of the persistent hand-placed kernels (forward, dQ): which items a workgroup walks, and the compaction of
// its valid items into the LDS descriptor table the asm bodies read (tools/asmgen/worklist.py).
#pragma once
#include "sfa_common.hpp"

namespace sfa {

// Items are ranked by cost, longest first (query tiles from the END of the sequence first, every (batch, KV head, head
// set) group per tile).  Round j hands ranks [j G, (j + 1) G) to the G workgroups: forwards in even rounds, backwards in
// odd ones ("snake"), so that every workgroup's total is the same to within one round's spread - the static stand-in
// for the hardware's longest-first dispatch of one-item workgroups.  When G is a multiple of 8 the mirror stays inside
// the workgroup's XCD lane (w % 8; blocks are dealt round-robin over the 8 XCDs), so that an XCD keeps seeing the same
// (batch, KV head) groups: their K / V tiles meet in its L2.  Speed only: nothing depends on the placement.
__device__ __forceinline__ int64_t wl_rank(int j, int w, int G) {
    int pos = w;
    if (j & 1) pos = (G & 7) == 0 ? ((G >> 3) - 1 - (w >> 3)) * 8 + (w & 7) : G - 1 - w;
    return (int64_t)j * G + pos;
}

// index of this thread's item among the workgroup's valid ones (256 threads, 4 waves); *total = their number.
// `cnt`: 4 ints of LDS scratch.  Two workgroup barriers.
__device__ __forceinline__ int wl_compact(bool valid, int* cnt, int* total) {
    const unsigned long long m = __ballot(valid);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();                       // (the previous chunk's readers of cnt are done)
    if (lane == 0) cnt[wv] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = cnt[i];
        if (i < wv) base += c;
        tot += c;
    }
    *total = tot;
    return base + __popcll(m & ((1ull << lane) - 1ull));
}

// Ticketed dispatch: which work item a freshly started workgroup takes.  The hardware deals workgroups to the 8 XCDs
// round-robin, a fixed share each, and XCDs of one chip differ in speed by 5-8 % under these kernels (per-XCD life times
// in profiles/r03_stamps_wl_*.log): with the block id as work id the fast XCDs idle while the slowest finishes its
// share.  Here the n items are cut into 8 queues exactly as xcd_remap() cuts them (so an XCD keeps seeing the same
// (batch, KV head) groups while its own queue lasts); a workgroup takes the head of the queue of the XCD it really runs
// on (HW_REG_XCC_ID) and, once that is empty, of the next non-empty one.  The launch adds surplus workgroups (they
// find nothing and exit), so an XCD that finishes early still has workgroups to start.  `heads`: 8 counters, 32 dwords
// apart, zeroed before the launch.  Returns the item (an xcd_remap'ed id) or -1; one LDS int of scratch, one barrier.
__device__ __forceinline__ int wl_ticket(unsigned* heads, int n, int* scratch) {
    if (threadIdx.x == 0) {
        const int x = (int)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7;       // HW_REG_XCC_ID
        const int q8 = n >> 3, r8 = n & 7;
        int got = -1;
        for (int k = 0; k < 8 && got < 0; ++k) {
            const int y = (x + k) & 7;
            const int cnt = q8 + (y < r8 ? 1 : 0);
            if (cnt == 0) continue;
            const unsigned t = atomicAdd(heads + 32 * y, 1u);
            if (t < (unsigned)cnt) got = (y < r8 ? y * (q8 + 1) : r8 * (q8 + 1) + (y - r8) * q8) + (int)t;
        }
        *scratch = got;
    }
    __syncthreads();
    const int r = *scratch;
    __syncthreads();
    return r;
}

}  // namespace sfa
