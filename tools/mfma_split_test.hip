// How exact is an fp32 contraction carried by bf16 MFMAs on split operands?  (gfx950, one wavefront, C[32x32] = A[32xK] B[Kx32].)
//   a = h + m + l with h = bf16(a), m = bf16(a - h), l = bf16(a - h - m) (round to nearest even; the remainder is below 2^-26 |a|);
//   x9: all nine plane products; x6: hh, hm, mh, hl, lh, mm (dropped: ml, lm, ll <= 2^-26 of the product); x3: hh, hm, mh; x1: plain bf16.
// Compared against a float64 host sum and against v_mfma_f32_32x32x2_f32 on the unsplit operands.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_split_test.hip -o /tmp/mfma_split_test && /tmp/mfma_split_test
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline void split3(float a, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)a;
    const float r1 = a - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// mode 0: fp32 MFMA; 1: bf16; 3 / 6 / 9: split products; 16: x6 with the small products in an accumulator of their own
__global__ void contract(const float* A, const float* B, float* C, int K, int mode) {
    const int lane = threadIdx.x, rc = lane & 31, g = lane >> 5;
    f32x16 acc = {0}, small = {0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rc * K + k + g], B[(k + g) * 32 + rc], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 a[3], b[3];
            for (int e = 0; e < 8; ++e) {
                __bf16 h, m, l;
                split3(A[rc * K + k + g * 8 + e], h, m, l);
                a[0][e] = h; a[1][e] = m; a[2][e] = l;
                split3(B[(k + g * 8 + e) * 32 + rc], h, m, l);
                b[0][e] = h; b[1][e] = m; b[2][e] = l;
            }
            auto mm = [&](f32x16& d, int i, int j) { d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], d, 0, 0, 0); };
            if (mode == 16) {
                mm(small, 2, 0); mm(small, 0, 2); mm(small, 1, 1); mm(small, 1, 0); mm(small, 0, 1);
                mm(acc, 0, 0);
            } else {
                if (mode >= 9) { mm(acc, 2, 2); mm(acc, 2, 1); mm(acc, 1, 2); }
                if (mode >= 6) { mm(acc, 2, 0); mm(acc, 0, 2); mm(acc, 1, 1); }
                if (mode >= 3) { mm(acc, 1, 0); mm(acc, 0, 1); }
                mm(acc, 0, 0);
            }
        }
        if (mode == 16) acc += small;
    }
    for (int i = 0; i < 16; ++i) C[((i / 4) * 8 + g * 4 + (i % 4)) * 32 + rc] = acc[i];
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return std::sqrt(-2.0 * std::log(urand())) * std::cos(6.283185307179586 * urand()); }

int main() {
    const int Ks[] = {576, 4608};
    for (int dist = 0; dist < 3; ++dist)
        for (int K : Ks) {
            std::vector<float> A(32 * K), B(K * 32), C(32 * 32);
            srand(7 + dist);
            for (auto& v : A) v = dist == 0 ? (float)nrand() : dist == 1 ? (float)std::fmax(nrand(), 0.0) : (float)(nrand() * std::exp(3.0 * nrand()));
            for (auto& v : B) v = (float)(nrand() * (dist == 2 ? std::exp(3.0 * nrand()) : 0.05));
            std::vector<double> R(32 * 32, 0.0), S(32 * 32, 0.0);
            for (int r = 0; r < 32; ++r)
                for (int c = 0; c < 32; ++c)
                    for (int k = 0; k < K; ++k) {
                        R[r * 32 + c] += (double)A[r * K + k] * B[k * 32 + c];
                        S[r * 32 + c] += std::fabs((double)A[r * K + k] * B[k * 32 + c]);
                    }
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            printf("operands %s, K = %d   (error relative to sum |a b|, in units of 2^-24)\n", dist == 0 ? "normal" : dist == 1 ? "relu(normal)" : "log-normal scales", K);
            const int modes[] = {0, 9, 6, 16, 3, 1};
            for (int mode : modes) {
                hipLaunchKernelGGL(contract, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
                hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
                double mx = 0, sq = 0;
                for (int i = 0; i < 32 * 32; ++i) {
                    const double e = std::fabs(C[i] - R[i]) / S[i] * 16777216.0;
                    mx = std::fmax(mx, e);
                    sq += e * e;
                }
                printf("   %-34s max %10.3f   rms %10.3f\n", mode == 0 ? "v_mfma_f32_32x32x2_f32" : mode == 1 ? "bf16" : mode == 3 ? "bf16 x3" : mode == 6 ? "bf16 x6" :
                       mode == 16 ? "bf16 x6, small terms apart" : "bf16 x9", mx, std::sqrt(sq / 1024));
            }
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    return 0;
}
