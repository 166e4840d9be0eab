// Does v_mfma_f64_16x16x4_f64 accumulate its 4 k-products as a sequential fma chain (k ascending) on top of C?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
__global__ void k(const double* A, const double* B, const double* C, double* D) {
    // layout (cdna guide): A 16x4: lane l holds A[l%16][l/16]; B 4x16: lane l holds B[l/16][l%16];
    // C/D 16x16: lane l holds 4 values D[4*(l/16)+r][l%16], r=0..3
    const int l = threadIdx.x;
    double a = A[(l % 16) * 4 + (l / 16)];
    double b = B[(l / 16) * 16 + (l % 16)];
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[((l / 16) + 4 * r) * 16 + (l % 16)];
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l / 16) + 4 * r) * 16 + (l % 16)] = d[r];
}
int main() {
    double hA[64], hB[64], hC[256], hD[256];
    srand(1);
    auto rnd = []() { return (rand() / (double)RAND_MAX - 0.5) * pow(2.0, (rand() % 40) - 20); };
    int bad_chain = 0, bad_rev = 0, bad_layout = 0;
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
    for (int trial = 0; trial < 200; ++trial) {
        for (double& v : hA) v = rnd(); for (double& v : hB) v = rnd(); for (double& v : hC) v = rnd();
        hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
        hipMemcpy(dC, hC, 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double c1 = hC[i * 16 + j], c2 = c1, ex = 0;
            for (int kk = 0; kk < 4; ++kk) c1 = fma(hA[i * 4 + kk], hB[kk * 16 + j], c1);
            for (int kk = 3; kk >= 0; --kk) c2 = fma(hA[i * 4 + kk], hB[kk * 16 + j], c2);
            long double e = hC[i * 16 + j];
            for (int kk = 0; kk < 4; ++kk) e += (long double)hA[i * 4 + kk] * hB[kk * 16 + j];
            ex = (double)e;
            if (hD[i * 16 + j] != c1) ++bad_chain;
            if (hD[i * 16 + j] != c2) ++bad_rev;
            if (fabs(hD[i * 16 + j] - ex) > 1e-9 * fabs(ex) + 1e-30) ++bad_layout;
        }
    }
    printf("mismatch vs ascending fma chain: %d ; vs descending chain: %d ; gross (layout) errors: %d (of %d)\n", bad_chain, bad_rev, bad_layout, 200 * 256);
    return 0;
}
