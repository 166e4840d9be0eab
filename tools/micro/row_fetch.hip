// Micro-measurement: what one workgroup of 1024 threads (one CU) pays for a 128 KB row, by where the row is (HBM, memory-side
// cache, L2), by how many rows it keeps in flight, by how many CUs do the same at once.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/row_fetch tools/micro/row_fetch.hip; gpurun -- tools/micro/row_fetch   (profiles/r05_row_fetch.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double dv2 __attribute__((ext_vector_type(2)));
constexpr int ROW = 16384;   // doubles

template <int DEPTH, bool NT>
__global__ void __launch_bounds__(1024) fetch(const double *base, long row_stride_rows, int rows, double *sink, int wg_rows_apart) {
    const int t = threadIdx.x;
    const unsigned sl = ((t >> 6) << 9) + (t & 63);
    const dv2 *p = reinterpret_cast<const dv2 *>(base + (size_t)blockIdx.x * wg_rows_apart * ROW) + sl;
    double acc = 0.0;
    for (int r = 0; r < rows; r += DEPTH) {
        dv2 v[DEPTH][8];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const dv2 *a = p + (size_t)(r + d) * row_stride_rows * (ROW / 2) + q * 64;
                v[d][q] = NT ? __builtin_nontemporal_load(a) : *a;
            }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[d][q].x + v[d][q].y;
        __syncthreads();
    }
    if (acc == 12345.678) sink[t] = acc;
}

template <int DEPTH, bool NT>
float run(const double *buf, int wgs, int rows, int apart, double *sink, int reps = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(a);
        hipLaunchKernelGGL((fetch<DEPTH, NT>), dim3(wgs), dim3(1024), 0, 0, buf, 1, rows, sink, apart);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = std::min(best, ms);
    }
    return best;
}

int main() {
    const size_t total_rows = 65536;   // 8.6 GB
    double *buf, *sink;
    CK(hipMalloc(&buf, total_rows * ROW * sizeof(double)));
    CK(hipMalloc(&sink, 1024 * sizeof(double)));
    CK(hipMemset(buf, 0, total_rows * ROW * sizeof(double)));
    CK(hipDeviceSynchronize());
    printf("rows of 128 KB, one 1024-thread workgroup per CU; us per row per workgroup\n");
    for (int wgs : {1, 8, 32, 64, 128, 256}) {
        const int rows = 128;
        // HBM: every workgroup walks its own 128 consecutive rows, far from the others; first touch after a flush of caches by
        // reading 1 GB elsewhere is not attempted -- the buffer (8.6 GB) exceeds every cache, reps walk the same rows, so rep 1 = HBM
        // and later reps = whatever cache holds 128 KB * 128 * wgs
        float cold1 = 0;
        {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            // evict: stream 2 GB
            hipLaunchKernelGGL((fetch<1, false>), dim3(256), dim3(1024), 0, 0, buf + (size_t)40000 * ROW, 1, 64, sink, 64);
            hipEventRecord(a);
            hipLaunchKernelGGL((fetch<1, false>), dim3(wgs), dim3(1024), 0, 0, buf, 1, rows, sink, 128);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&cold1, a, b);
        }
        const float warm1 = run<1, false>(buf, wgs, rows, 128, sink);
        const float warm2 = run<2, false>(buf, wgs, rows, 128, sink);
        const float warm1nt = run<1, true>(buf, wgs, rows, 128, sink);
        // L2-resident: every workgroup re-reads the same 4 rows
        float l2;
        {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL((fetch<1, false>), dim3(wgs), dim3(1024), 0, 0, buf, 0, rows, sink, 0);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&l2, a, b);
        }
        printf("wgs %3d: cold(HBM) %.2f  re-read(depth1) %.2f  depth2 %.2f  nt %.2f  same-row(L2) %.2f   [set = %.0f MB]\n", wgs,
               1e3 * cold1 / rows, 1e3 * warm1 / rows, 1e3 * warm2 / rows, 1e3 * warm1nt / rows, 1e3 * l2 / rows,
               wgs * rows * 0.131072);
    }
    return 0;
}
