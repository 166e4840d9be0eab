// Micro-measurement: the rate v_mfma_f64_16x16x4_f64 sustains on this card, against the 78.6 TFLOP/s the guide derives from the
// clock (256 CUs x 4 SIMDs x 2048 flops / 64 cycles x 2.4 GHz). Register-only operands, no memory traffic: what a transform
// kernel could reach if it did nothing but issue matrix instructions. Shapes: waves per SIMD (1, 2, 4) x independent
// accumulators per wave (1, 2, 4, 8), and the same with a ds_read_b64 pair per instruction (the operands of h2d_kloop).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_f64_peak tools/micro/mfma_f64_peak.hip; gpurun -- tools/micro/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double d4_t __attribute__((ext_vector_type(4)));

// clk[0] += shader-clock ticks, clk[1] += 100 MHz ticks of wave 0 of every workgroup: their ratio is the clock the loop ran at
// MEM: every round also loads 16 B per lane from a large buffer (one K step's worth of operand traffic per ~16 matrix instructions)
template <int ACC, bool LDS, bool MEM = false>
__global__ void __launch_bounds__(256) spin(int iters, double *sink, double a0, double b0, unsigned long long *clk = nullptr,
                                            const double2 *stream = nullptr, size_t stream_len = 0) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double2 mv = make_double2(0.0, 0.0);
    size_t mpos = ((size_t)blockIdx.x * 256 + threadIdx.x) % (stream_len ? stream_len : 1);
    __shared__ double s[2 * 32 * 80];
    const int lane = threadIdx.x & 63;
    if (LDS) {
        for (int i = threadIdx.x; i < 2 * 32 * 80; i += 256) s[i] = a0 + i * 1e-9;
        __syncthreads();
    }
    d4_t acc[ACC];
#pragma unroll
    for (int k = 0; k < ACC; ++k) acc[k] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double a = a0 + lane * 1e-6, b = b0 - lane * 1e-6;
    for (int it0 = 0; it0 < iters; it0 += 8) {   // (eight rounds per trip: the compiler moves the accumulators between the two register
        if (MEM) {                               //  files at the loop's edge)
            const double2 q = stream[mpos];
            mv.x += q.x; mv.y += q.y;
            mpos += 262144; if (mpos >= stream_len) mpos -= stream_len;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
        const int it = it0 + u;
#pragma unroll
        for (int k = 0; k < ACC; ++k) {
            if (LDS) {
                a = s[((it + k) & 31) * 80 + (lane & 15) + (lane >> 4) * 80 * 0 + (k & 1) * 16];
                b = s[32 * 80 + ((it + k) & 31) * 80 + (lane & 15) + (k >> 1 & 1) * 16];
            }
            acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
        }
        }
    }
    double r = 0.0;
#pragma unroll
    for (int k = 0; k < ACC; ++k) r += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    if (r == 12345.678 || mv.x == 12345.678) sink[threadIdx.x] = r + mv.y;
    if (clk && threadIdx.x == 0) {
        atomicAdd(clk, __builtin_amdgcn_s_memtime() - c0);
        atomicAdd(clk + 1, __builtin_amdgcn_s_memrealtime() - r0);
    }
}

template <int ACC, bool LDS>
int run(int wg_per_cu, double *sink) {
    const int iters = 4096 / ACC * 4, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((spin<ACC, LDS>), dim3(grid), dim3(256), 0, 0, iters, sink, 1.0, 2.0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double flops = (double)grid * 4 * iters * ACC * 2048.0;
    printf("waves/SIMD %d  accumulators %d  %s  %8.3f ms  %6.1f TFLOP/s  %.3f of 78.6\n", wg_per_cu, ACC, LDS ? "lds operands" : "reg operands",
           best, flops / best * 1e-9, flops / best * 1e-9 / 78.6);
    return 0;
}

int main() {
    double *sink;
    CK(hipMalloc(&sink, 4096));
    for (int w : {1, 2, 4}) {
        if (run<1, false>(w, sink) || run<2, false>(w, sink) || run<4, false>(w, sink) || run<8, false>(w, sink)) return 1;
    }
    for (int w : {1, 2, 4}) {
        if (run<4, true>(w, sink) || run<8, true>(w, sink)) return 1;
    }
    {   // the clock the matrix loop runs at, alone and with operand traffic beside it
        unsigned long long *clk;
        CK(hipMalloc(&clk, 16));
        double2 *stream;
        const size_t slen = (size_t)1 << 28;   // 4 GB of double2
        CK(hipMalloc(&stream, slen * sizeof(double2)));
        CK(hipMemset(stream, 0, slen * sizeof(double2)));
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipMemset(clk, 0, 16));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            for (int k = 0; k < 20; ++k) {
                if (mode == 0) hipLaunchKernelGGL((spin<4, false, false>), dim3(1024), dim3(256), 0, 0, 4096, sink, 1.0, 2.0, clk, stream, slen);
                else if (mode == 1) hipLaunchKernelGGL((spin<4, true, false>), dim3(1024), dim3(256), 0, 0, 4096, sink, 1.0, 2.0, clk, stream, slen);
                else hipLaunchKernelGGL((spin<4, true, true>), dim3(1024), dim3(256), 0, 0, 4096, sink, 1.0, 2.0, clk, stream, slen);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[2];
            CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double flops = 20.0 * 1024 * 4 * 4096 * 4 * 2048.0;
            printf("%-44s %6.1f TFLOP/s  %.3f of 78.6   shader clock %.0f MHz\n",
                   mode == 0 ? "clock: register operands" : mode == 1 ? "clock: LDS operands" : "clock: LDS operands + 16 B/lane per 32 MFMA from HBM",
                   flops / ms * 1e-9, flops / ms * 1e-9 / 78.6, 100.0 * (double)h[0] / (double)h[1]);
        }
    }
    // a long run: does the rate hold once the card has been at it for a second (clocks under a sustained FP64 matrix load)?
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int k = 0; k < 100; ++k) hipLaunchKernelGGL((spin<4, false>), dim3(1024), dim3(256), 0, 0, 4096, sink, 1.0, 2.0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = 100.0 * 1024 * 4 * 4096 * 4 * 2048.0;
        printf("sustained (100 launches, %.0f ms): %6.1f TFLOP/s  %.3f of 78.6\n", ms, flops / ms * 1e-9, flops / ms * 1e-9 / 78.6);
    }
    return 0;
}
