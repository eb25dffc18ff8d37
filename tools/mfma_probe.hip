// mfma_probe.hip -- prints where v_mfma_f32_32x32x2_f32 puts D[i][j] (lane, register), measured, and the
// largest difference between the MFMA dot products and the lane-ordered fp32 sums the distance kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void probe(const float *A /*32 x K row-major*/, const float *B /*32 x K row-major*/, int K, float *out /*64 x 16*/)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    floatx16 acc = {0};
    for (int k0 = 0; k0 < K; k0 += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k0 + h], B[r * K + k0 + h], acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) out[lane * 16 + v] = acc[v];
}
int main()
{
    const int K = 768;
    std::vector<float> A(32 * K), B(32 * K), out(64 * 16);
    srand(7);
    for (auto &x : A) x = rand() / (float)RAND_MAX;
    for (auto &x : B) x = rand() / (float)RAND_MAX;
    for (int i = 0; i < 32; ++i) { // unit rows
        double na = 0, nb = 0;
        for (int k = 0; k < K; ++k) { na += (double)A[i * K + k] * A[i * K + k]; nb += (double)B[i * K + k] * B[i * K + k]; }
        for (int k = 0; k < K; ++k) { A[i * K + k] /= (float)sqrt(na); B[i * K + k] /= (float)sqrt(nb); }
    }
    float *dA, *dB, *dO;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, out.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, K, dO);
    hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost);
    // guess: D[i][j] with j = lane % 32, i = 8 * (v / 4) + 4 * (lane / 32) + v % 4
    double worst_guess = 0, worst_lane_order = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int v = 0; v < 16; ++v) {
            const int j = lane % 32, i = 8 * (v / 4) + 4 * (lane / 32) + v % 4;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)A[i * K + k] * (double)B[j * K + k];
            worst_guess = fmax(worst_guess, fabs(ref - out[lane * 16 + v]));
            // the kernels' order: 8 partial sums, element k -> partial k % 8, mul then add, tree ((p0+p4)+(p2+p6))+((p1+p5)+(p3+p7))
            float p[8] = {0};
            for (int k = 0; k < K; ++k) { float pr = A[i * K + k] * B[j * K + k]; p[k % 8] = p[k % 8] + pr; }
            float u0 = p[0] + p[4], u1 = p[1] + p[5], u2 = p[2] + p[6], u3 = p[3] + p[7];
            float s = (u0 + u2) + (u1 + u3);
            worst_lane_order = fmax(worst_lane_order, fabs((double)s - out[lane * 16 + v]));
        }
    printf("layout guess j=lane%%32, i=8*(v/4)+4*(lane/32)+v%%4: max |D - float64 dot| = %.3e (%s); max |MFMA - lane-ordered fp32| = %.3e (K = %d, unit rows)\n",
           worst_guess, worst_guess < 1e-4 ? "CONFIRMED" : "WRONG", worst_lane_order, K);
    return worst_guess < 1e-4 ? 0 : 1;
}
