// Hardware-layout probes for gfx950 (run on the GPU box): checks the lane maps the MFMA kernels rely on.
//  1. v_mfma_f32_32x32x16_bf16 A/B/C maps        2. ds_read_b64_tr_b16 gather map
//  3. v_permlane32_swap semantics                 4. accumulator-as-B-operand k permutation
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

__global__ void k_mfma(const unsigned short* A, const unsigned short* B, float* C) {
    // A [32][16], B [16][32] bf16 row-major
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[r * 16 + 8 * h + j]; b[j] = B[(8 * h + j) * 32 + r]; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}

__global__ void k_tr(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short M[16][64];   // 128-byte rows
    const int l = threadIdx.x;
    for (int i = l; i < 16 * 64; i += 64) M[i / 64][i % 64] = (unsigned short)(((i / 64) << 8) | (i % 64));
    __syncthreads();
    // group g = l>>4 reads block rows 4g..4g+3 (q = (l&15)>>2), cols 16..31 (p = l&3 -> cols 16+4p..)
    const int g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    const unsigned addr = (unsigned)(size_t)&M[4 * g + q][16 + 4 * p];
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}

__global__ void k_swap(unsigned* out) {
    const int l = threadIdx.x;
    unsigned a = 1000 + l, b = 2000 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l * 2] = r[0];
    out[l * 2 + 1] = r[1];
}

// X = first product (32x32 f32 acc, column on lane), then Y = A2 * X using X as B operand (k = X row index)
__global__ void k_chain(const unsigned short* A1, const unsigned short* B1, const unsigned short* A2, float* Y) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A1[r * 16 + 8 * h + j]; b[j] = B1[(8 * h + j) * 32 + r]; }
    f32x16 x = {0};
    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);   // X [32][32] small ints, exact in bf16
    f32x16 y = {0};
    for (int s = 0; s < 2; ++s) {
        bf16x8 xb, a2;
        for (int j = 0; j < 8; ++j) {
            float f = x[8 * s + j];
            unsigned u = __float_as_uint(f);
            xb[j] = (short)(u >> 16);
            const int krow = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);   // documented k permutation
            a2[j] = A2[r * 32 + krow];                                   // A2 [32][32]
        }
        y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, y, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) Y[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = y[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    int fails = 0;
    {   // 1. mfma maps
        std::vector<unsigned short> A(32 * 16), B(16 * 32);
        std::vector<float> Af(32 * 16), Bf(16 * 32), C(32 * 32), Cr(32 * 32, 0.f);
        for (int i = 0; i < 32 * 16; ++i) { Af[i] = (float)((i * 7 + 3) % 13 - 6); A[i] = f2bf(Af[i]); }
        for (int i = 0; i < 16 * 32; ++i) { Bf[i] = (float)((i * 5 + 1) % 11 - 5); B[i] = f2bf(Bf[i]); }
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) Cr[i * 32 + j] += Af[i * 16 + k] * Bf[k * 32 + j];
        unsigned short *dA, *dB; float* dC;
        CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, C.size() * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        k_mfma<<<1, 64>>>(dA, dB, dC);
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 1024; ++i) bad += C[i] != Cr[i];
        printf("probe1 mfma_32x32x16 A/B/C maps: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    }
    {   // 2. tr read
        std::vector<unsigned short> out(256);
        unsigned short* d; CK(hipMalloc(&d, 512));
        k_tr<<<1, 64>>>(d);
        CK(hipMemcpy(out.data(), d, 512, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
            const int g = l >> 4, i = l & 15;
            const unsigned short want = (unsigned short)(((4 * g + e) << 8) | (16 + i));   // row 4g+e, col 16+i
            bad += out[l * 4 + e] != want;
        }
        printf("probe2 ds_read_b64_tr_b16 (lane i gets column i of rows q=0..3): %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
        if (bad) for (int l = 0; l < 64; l += 1) printf("  lane %2d: %04x %04x %04x %04x\n", l, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
        fails += bad != 0;
    }
    {   // 3. permlane32_swap
        std::vector<unsigned> out(128);
        unsigned* d; CK(hipMalloc(&d, 512));
        k_swap<<<1, 64>>>(d);
        CK(hipMemcpy(out.data(), d, 512, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            const unsigned w0 = l < 32 ? 1000 + l : 2000 + (l - 32);   // new vdst: low half keeps a, high half gets b[l-32]
            const unsigned w1 = l < 32 ? 1000 + l + 32 : 2000 + l;     // new src : low half gets a[l+32], high half keeps b
            bad += out[l * 2] != w0 || out[l * 2 + 1] != w1;
        }
        printf("probe3 permlane32_swap: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
        if (bad) for (int l = 0; l < 64; l += 8) printf("  lane %2d: r0=%u r1=%u\n", l, out[l * 2], out[l * 2 + 1]);
        fails += bad != 0;
    }
    {   // 4. chain
        std::vector<unsigned short> A1(32 * 16), B1(16 * 32), A2(32 * 32);
        std::vector<float> A1f(32 * 16), B1f(16 * 32), A2f(32 * 32), X(1024, 0.f), Yr(1024, 0.f), Y(1024);
        for (int i = 0; i < 512; ++i) { A1f[i] = (float)((i * 3 + 1) % 5 - 2); A1[i] = f2bf(A1f[i]); B1f[i] = (float)((i * 7 + 2) % 3 - 1); B1[i] = f2bf(B1f[i]); }
        for (int i = 0; i < 1024; ++i) { A2f[i] = (float)((i * 11 + 5) % 7 - 3); A2[i] = f2bf(A2f[i]); }
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) X[i * 32 + j] += A1f[i * 16 + k] * B1f[k * 32 + j];
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 32; ++k) Yr[i * 32 + j] += A2f[i * 32 + k] * X[k * 32 + j];
        unsigned short *d1, *d2, *d3; float* dY;
        CK(hipMalloc(&d1, 1024)); CK(hipMalloc(&d2, 1024)); CK(hipMalloc(&d3, 2048)); CK(hipMalloc(&dY, 4096));
        CK(hipMemcpy(d1, A1.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(d2, B1.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(d3, A2.data(), 2048, hipMemcpyHostToDevice));
        k_chain<<<1, 64>>>(d1, d2, d3, dY);
        CK(hipMemcpy(Y.data(), dY, 4096, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 1024; ++i) bad += Y[i] != Yr[i];
        printf("probe4 accumulator-as-B-operand k permutation: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    }
    printf("probes: %s\n", fails ? "SOME FAILED" : "ALL PASS");
    return fails ? 1 : 0;
}
