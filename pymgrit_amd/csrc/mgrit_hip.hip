// mgrit_hip.hip -- MI355X (gfx950 / CDNA4) MGRIT relaxation engine: HIP kernels + the C ABI of include/mgrit_hip.h.
//
// Replaces the per-time-point Python loops of the reference's hot path (src/pymgrit/core/mgrit.py:292-549,715-726)
// and the per-step SuperLU solves of heat/heat_1d.py:198-217 and advection/advection_1d.py:129-143.
//
// Data layout: every level keeps its time-point states in one row-major float64 slab [n_local_points][ld] in HBM
// (x contiguous -> every global access below is a 16-byte-per-lane coalesced stream).
// Execution model: ONE workgroup per run of consecutive time points (an F-interval, a C-point, or the coarsest-level
// chain). The workgroup keeps the whole state vector in registers (16 consecutive x per lane, 64 lanes per wave,
// up to 16 waves = 16384 x), converts between the coalesced global layout and the per-lane blocked layout through a
// per-wave padded LDS tile (no workgroup barrier), and applies Phi as two constant-coefficient first-order
// recurrences (forward, backward) + a rank-one correction -- each recurrence is a chunked scan: lane-local FMA chain,
// Kogge-Stone over the 64 lanes of a wave via cross-lane shuffles, serial carry across waves through LDS.
// The arithmetic (operation order, FMA placement, reduction trees) is specified in DESIGN.md section 3 and must
// match oracle/mgrit_oracle.c variant 1 bit for bit: compile with -ffp-contract=off; every FMA is explicit.
//
// gfx950 only. No CUDA paths, no fallbacks: every entry point fails when no HIP device is usable.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mgrit_hip.h"

namespace {

constexpr int E = MGRIT_HIP_E;          // elements per lane
constexpr int LANES = 64;               // lanes per wave
constexpr int GROUP = E * LANES;        // elements per wave
constexpr int WAVE_TILE_BYTES = LANES * (E * 8 + 16);  // 144-byte padded lane rows: conflict-free b128 access
constexpr int MAX_G = MGRIT_HIP_MAX_N / GROUP;         // 16 waves

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t err__ = (expr);                                                                 \
        if (err__ != hipSuccess) return fail(MGRIT_HIP_EHIP, "%s: %s", #expr, hipGetErrorString(err__)); \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------
// Coefficient set: everything Phi needs for one distinct dt on one level
// ---------------------------------------------------------------------------------------------------------------
struct CSet {
    double rho, ik, scal, gc;
    double pw[E + 1];   // rho^k
    double sc[6];       // rho^(E*2^s)
    double lp[LANES];   // rho^(E*l)
};

struct LevelDev {
    double *u, *v, *g;
    const int32_t *cidx;  // [n_pts] coefficient set of the step (i-1 -> i)
    const double *dt;     // [n_pts]
    const double *tau;    // [K][n_pts]
    const double *sT;     // [K][E][T]   forcing space factors, lane-transposed
    const CSet *cs;       // [n_csets]
    const double *tabT;   // [n_csets][E][T] rank-one correction table, lane-transposed
    int n, ld, T, n_pts, K, kind;
};

// ---------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// coalesced global row -> blocked registers (lane owns x[16t .. 16t+15]) through this wave's padded LDS tile
__device__ __forceinline__ void load_row(const double *__restrict__ row, int n, int ld, char *tile, int lane, int wave,
                                         double (&x)[E]) {
    const int base = wave * GROUP;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = base + 128 * q + 2 * lane;
        double2 v = make_double2(0.0, 0.0);
        if (e < ld) v = *reinterpret_cast<const double2 *>(row + e);
        if (e >= n) v.x = 0.0;
        if (e + 1 >= n) v.y = 0.0;
        const int o = 8 * q + (lane >> 3), k = 2 * (lane & 7);
        *reinterpret_cast<double2 *>(tile + o * 144 + k * 8) = v;
    }
    wave_sync();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const double2 v = *reinterpret_cast<const double2 *>(tile + lane * 144 + 16 * q);
        x[2 * q] = v.x;
        x[2 * q + 1] = v.y;
    }
    wave_sync();
}

// blocked registers -> coalesced global row (padding columns [n, ld) are written as zero)
__device__ __forceinline__ void store_row(double *__restrict__ row, int n, int ld, char *tile, int lane, int wave,
                                          const double (&x)[E]) {
    const int base = wave * GROUP;
#pragma unroll
    for (int q = 0; q < 8; ++q)
        *reinterpret_cast<double2 *>(tile + lane * 144 + 16 * q) = make_double2(x[2 * q], x[2 * q + 1]);
    wave_sync();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = base + 128 * q + 2 * lane;
        const int o = 8 * q + (lane >> 3), k = 2 * (lane & 7);
        double2 v = *reinterpret_cast<const double2 *>(tile + o * 144 + k * 8);
        if (e >= n) v.x = 0.0;
        if (e + 1 >= n) v.y = 0.0;
        if (e < ld) *reinterpret_cast<double2 *>(row + e) = v;
    }
    wave_sync();
}

// forward chunked scan  y_j = rho*y_{j-1} + d_j   -- two workgroup barriers are the caller's:
// this routine ends with x holding y; tot[] is an LDS array of MAX_G doubles.
__device__ __forceinline__ void scan_fwd(double (&x)[E], const CSet &c, double *tot, int lane, int wave) {
#pragma unroll
    for (int k = 1; k < E; ++k) x[k] = fma(c.rho, x[k - 1], x[k]);
    double a = x[E - 1];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const double v = __shfl_up(a, 1u << s);
        if (lane >= (1 << s)) a = fma(c.sc[s], v, a);
    }
    if (lane == LANES - 1) tot[wave] = a;
    __syncthreads();
    double carry = 0.0;
    for (int g = 0; g < wave; ++g) carry = fma(c.gc, carry, tot[g]);
    double prev = __shfl_up(a, 1u);
    if (lane == 0) prev = 0.0;
    const double cin = fma(c.lp[lane], carry, prev);
#pragma unroll
    for (int k = 0; k < E; ++k) x[k] = fma(c.pw[k + 1], cin, x[k]);
}

// backward chunked scan  z_j = rho*z_{j+1} + y_j
__device__ __forceinline__ void scan_bwd(double (&x)[E], const CSet &c, double *tot, int lane, int wave, int G) {
#pragma unroll
    for (int k = E - 2; k >= 0; --k) x[k] = fma(c.rho, x[k + 1], x[k]);
    double a = x[0];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const double v = __shfl_down(a, 1u << s);
        if (lane + (1 << s) < LANES) a = fma(c.sc[s], v, a);
    }
    if (lane == 0) tot[wave] = a;
    __syncthreads();
    double carry = 0.0;
    for (int g = G - 1; g > wave; --g) carry = fma(c.gc, carry, tot[g]);
    double next = __shfl_down(a, 1u);
    if (lane == LANES - 1) next = 0.0;
    const double cin = fma(c.lp[LANES - 1 - lane], carry, next);
#pragma unroll
    for (int k = 0; k < E; ++k) x[k] = fma(c.pw[E - k], cin, x[k]);
}

struct Smem {
    char *tile;     // this wave's transposition tile
    double *totF;   // [MAX_G]
    double *totB;   // [MAX_G]
    double *bc;     // [2] broadcast scalars
};

__device__ __forceinline__ Smem carve_smem(char *base, int G, int wave) {
    Smem s;
    s.tile = base + wave * WAVE_TILE_BYTES;
    double *tail = reinterpret_cast<double *>(base + G * WAVE_TILE_BYTES);
    s.totF = tail;
    s.totB = tail + MAX_G;
    s.bc = tail + 2 * MAX_G;
    return s;
}

// x <- Phi(x) for the step (i-1 -> i) of level L.  heat_1d.py:198-217 / advection_1d.py:129-143.
template <int KIND>
__device__ __forceinline__ void phi_apply(double (&x)[E], const LevelDev &L, int i, const Smem &sm, int t, int lane,
                                          int wave, int G) {
    const CSet &c = L.cs[L.cidx[i]];
    const double *tab = L.tabT + (size_t)L.cidx[i] * E * L.T;
    const int j0 = t * E;
    if (KIND == MGRIT_HIP_STEPPER_HEAT1D) {
        if (L.K > 0) {
            const double dt = L.dt[i];
#pragma unroll
            for (int k = 0; k < E; ++k) {
                double f = L.sT[k * L.T + t] * L.tau[i];
                for (int kk = 1; kk < L.K; ++kk) f = f + L.sT[(size_t)(kk * E + k) * L.T + t] * L.tau[(size_t)kk * L.n_pts + i];
                x[k] = x[k] + f * dt;
            }
        }
        scan_fwd(x, c, sm.totF, lane, wave);
#pragma unroll
        for (int k = 0; k < E; ++k)
            if (j0 + k >= L.n) x[k] = 0.0;
        scan_bwd(x, c, sm.totB, lane, wave, G);
        if (t == 0) sm.bc[0] = x[0] * c.ik;
        __syncthreads();
        const double z0 = sm.bc[0];
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = fma(-z0, tab[k * L.T + t], x[k] * c.ik);
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = x[k] * c.ik;
        scan_fwd(x, c, sm.totF, lane, wave);
        const int jl = L.n - 1;
        if (t == jl / E) {
            double y = 0.0;
#pragma unroll
            for (int k = 0; k < E; ++k)
                if (k == jl % E) y = x[k];
            sm.bc[0] = y * c.scal;
        }
        __syncthreads();
        const double xl = sm.bc[0];
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = (j0 + k < L.n) ? fma(tab[k * L.T + t], xl, x[k]) : 0.0;
        __syncthreads();  // protects totF / bc reuse by the next step (heat has 3 barriers per step, advection 2 + this)
    }
}

// sum of squares of the lane-blocked vector r with the spec's reduction tree; result valid in thread 0
__device__ __forceinline__ double block_sumsq(const double (&r)[E], const Smem &sm, int t, int lane, int wave, int G) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < E; ++k) acc = fma(r[k], r[k], acc);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = acc + __shfl_xor(acc, off);
    __syncthreads();
    if (lane == 0) sm.totF[wave] = acc;
    __syncthreads();
    double tot = 0.0;
    if (t == 0)
        for (int g = 0; g < G; ++g) tot = tot + sm.totF[g];
    return tot;
}

// ---------------------------------------------------------------------------------------------------------------
// kernels (one workgroup per run / pair)
// ---------------------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) char smem_raw[];

// f_relax / c_relax / forward_solve (mgrit.py:292-370,459-486)
template <int KIND>
__global__ void __launch_bounds__(1024) relax_kernel(LevelDev L, const int32_t *__restrict__ run_start,
                                                     const int32_t *__restrict__ run_len, int use_g, int mode,
                                                     double w, double w1) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, G = blockDim.x >> 6;
    const Smem sm = carve_smem(smem_raw, G, wave);
    const int start = run_start[blockIdx.x], len = run_len[blockIdx.x];
    double x[E];
    load_row(L.u + (size_t)(start - 1) * L.ld, L.n, L.ld, sm.tile, lane, wave, x);
    for (int i = start; i < start + len; ++i) {
        phi_apply<KIND>(x, L, i, sm, t, lane, wave, G);
        if (use_g) {
            double gi[E];
            load_row(L.g + (size_t)i * L.ld, L.n, L.ld, sm.tile, lane, wave, gi);
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = gi[k] + x[k];
        }
        if (mode == MGRIT_HIP_RELAX_C && w != 1.0) {
            double uo[E];
            load_row(L.u + (size_t)i * L.ld, L.n, L.ld, sm.tile, lane, wave, uo);
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = x[k] * w + uo[k] * w1;
        }
        store_row(L.u + (size_t)i * L.ld, L.n, L.ld, sm.tile, lane, wave, x);
    }
}

// compute_residual (mgrit.py:387-413): out[run] = || Phi(u_{i-1}) - u_i ||^2
template <int KIND>
__global__ void __launch_bounds__(1024) residual_kernel(LevelDev L, const int32_t *__restrict__ run_start,
                                                        double *__restrict__ out) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, G = blockDim.x >> 6;
    const Smem sm = carve_smem(smem_raw, G, wave);
    const int i = run_start[blockIdx.x];
    double x[E], ui[E];
    load_row(L.u + (size_t)(i - 1) * L.ld, L.n, L.ld, sm.tile, lane, wave, x);
    phi_apply<KIND>(x, L, i, sm, t, lane, wave, G);
    load_row(L.u + (size_t)i * L.ld, L.n, L.ld, sm.tile, lane, wave, ui);
#pragma unroll
    for (int k = 0; k < E; ++k) x[k] = x[k] - ui[k];
    const double tot = block_sumsq(x, sm, t, lane, wave, G);
    if (t == 0) out[blockIdx.x] = tot;
}

// compute_jump (mgrit.py:372-385): out[run] = || u_i - prev_i ||^2
__global__ void __launch_bounds__(1024) jump_kernel(LevelDev L, const int32_t *__restrict__ run_start,
                                                    const double *__restrict__ prev, double *__restrict__ out) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, G = blockDim.x >> 6;
    const Smem sm = carve_smem(smem_raw, G, wave);
    const int i = run_start[blockIdx.x];
    double x[E], p[E];
    load_row(L.u + (size_t)i * L.ld, L.n, L.ld, sm.tile, lane, wave, x);
    load_row(prev + (size_t)i * L.ld, L.n, L.ld, sm.tile, lane, wave, p);
#pragma unroll
    for (int k = 0; k < E; ++k) x[k] = x[k] - p[k];
    const double tot = block_sumsq(x, sm, t, lane, wave, G);
    if (t == 0) out[blockIdx.x] = tot;
}

// fas_residual, fine half (mgrit.py:528-532 / 538-543): out_p = Phi_l(u_{i-1}) - u_i   or   (g_i - u_i) + Phi_l(u_{i-1})
template <int KIND>
__global__ void __launch_bounds__(1024) fas_fine_kernel(LevelDev L, const int32_t *__restrict__ fine_idx,
                                                        const int32_t *__restrict__ out_idx, double *__restrict__ out,
                                                        int out_ld, int use_g) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, G = blockDim.x >> 6;
    const Smem sm = carve_smem(smem_raw, G, wave);
    const int i = fine_idx[blockIdx.x];
    double x[E], ui[E];
    load_row(L.u + (size_t)(i - 1) * L.ld, L.n, L.ld, sm.tile, lane, wave, x);
    phi_apply<KIND>(x, L, i, sm, t, lane, wave, G);
    load_row(L.u + (size_t)i * L.ld, L.n, L.ld, sm.tile, lane, wave, ui);
    if (use_g) {
        double gi[E];
        load_row(L.g + (size_t)i * L.ld, L.n, L.ld, sm.tile, lane, wave, gi);
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = (gi[k] - ui[k]) + x[k];
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = x[k] - ui[k];
    }
    store_row(out + (size_t)out_idx[blockIdx.x] * out_ld, L.n, out_ld, sm.tile, lane, wave, x);
}

// fas_residual, coarse half (mgrit.py:533-536 / 544-547): g_j = (g_j + v_j) - Phi_{l+1}(v_{j-1})
template <int KIND>
__global__ void __launch_bounds__(1024) fas_coarse_kernel(LevelDev L, const int32_t *__restrict__ coarse_idx) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, G = blockDim.x >> 6;
    const Smem sm = carve_smem(smem_raw, G, wave);
    const int j = coarse_idx[blockIdx.x];
    double x[E], a[E], b[E];
    load_row(L.v + (size_t)(j - 1) * L.ld, L.n, L.ld, sm.tile, lane, wave, x);
    phi_apply<KIND>(x, L, j, sm, t, lane, wave, G);
    load_row(L.g + (size_t)j * L.ld, L.n, L.ld, sm.tile, lane, wave, a);
    load_row(L.v + (size_t)j * L.ld, L.n, L.ld, sm.tile, lane, wave, b);
#pragma unroll
    for (int k = 0; k < E; ++k) x[k] = (a[k] + b[k]) - x[k];
    store_row(L.g + (size_t)j * L.ld, L.n, L.ld, sm.tile, lane, wave, x);
}

// --- spatial transfer kernels (bandwidth-bound, elementwise). grid = (pairs, ceil(n_out/256)) -------------------
// restriction: dst row d_idx[p] (n_c) <- R(src row s_idx[p] (n_f)). kind 0 copy; kind 1 full weighting
// (examples/example_spatial_coarsening.py:33-55: sol[2i]*1/4 + sol[2i+1]*1/2 + sol[2i+2]*1/4).
__global__ void restrict_rows_kernel(const double *__restrict__ src, int src_ld, const int32_t *__restrict__ s_idx,
                                     double *__restrict__ dst, int dst_ld, const int32_t *__restrict__ d_idx, int n_c,
                                     int kind) {
    const int p = blockIdx.x, i = blockIdx.y * blockDim.x + threadIdx.x;
    if (i >= n_c) return;
    const double *f = src + (size_t)s_idx[p] * src_ld;
    double *c = dst + (size_t)d_idx[p] * dst_ld;
    if (kind == MGRIT_HIP_TRANSFER_COPY) c[i] = f[i];
    else c[i] = f[2 * i] * 1.0 / 4.0 + f[2 * i + 1] * 1.0 / 2.0 + f[2 * i + 2] * 1.0 / 4.0;
}

// interpolation value at fine index j of coarse vector e (examples/example_spatial_coarsening.py:58-82)
__device__ __forceinline__ double interp_at(const double *e, const double *e2, int n_c, int j, int kind) {
    // value of P(e - e2) (e2 may be null -> P(e))
    auto at = [&](int i) { return e2 ? e[i] - e2[i] : e[i]; };
    if (kind == MGRIT_HIP_TRANSFER_COPY) return at(j);
    if (j & 1) return 0.0 + at(j >> 1);
    const int i = j >> 1;
    double r = 0.0;
    if (i - 1 >= 0) r = r + 1.0 / 2.0 * at(i - 1);
    if (i < n_c) r = r + 1.0 / 2.0 * at(i);
    return r;
}

// mode 0: u^l_i = P(u^{l+1}_j)  (mgrit.py:562-563);  mode 1: u^l_i = u^l_i + P(u^{l+1}_j - v^{l+1}_j)  (mgrit.py:724-726)
__global__ void interp_rows_kernel(double *__restrict__ uf, int f_ld, const int32_t *__restrict__ f_idx,
                                   const double *__restrict__ uc, const double *__restrict__ vc, int c_ld,
                                   const int32_t *__restrict__ c_idx, int n_f, int n_c, int kind, int mode) {
    const int p = blockIdx.x, j = blockIdx.y * blockDim.x + threadIdx.x;
    if (j >= n_f) return;
    double *f = uf + (size_t)f_idx[p] * f_ld;
    const double *e = uc + (size_t)c_idx[p] * c_ld;
    if (mode == 0) f[j] = interp_at(e, nullptr, n_c, j, kind);
    else f[j] = f[j] + interp_at(e, vc + (size_t)c_idx[p] * c_ld, n_c, j, kind);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct RunList { int n = 0; int32_t *d_start = nullptr, *d_len = nullptr; };
struct PairList { int n = 0; int32_t *d_fine = nullptr, *d_coarse = nullptr, *d_iota = nullptr; };

struct Level {
    bool set = false;
    LevelDev dev{};
    int G = 0, n_csets = 0, transfer = MGRIT_HIP_TRANSFER_COPY;
    std::vector<void *> allocs;
    std::vector<RunList> runs;
    std::vector<PairList> pairs;
    double *scratch = nullptr;
    size_t scratch_rows = 0;
};

}  // namespace

struct mgrit_hip_engine {
    int n_levels = 0;
    hipStream_t stream = nullptr;
    std::vector<Level> L;
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
};

namespace {

void cset_powers(CSet &c, double rho) {
    c.rho = rho;
    c.pw[0] = 1.0;
    c.pw[1] = rho;
    for (int k = 2; k <= E; ++k) c.pw[k] = c.pw[k - 1] * rho;
    c.sc[0] = c.pw[E];
    for (int s = 1; s < 6; ++s) c.sc[s] = c.sc[s - 1] * c.sc[s - 1];
    c.gc = c.sc[5] * c.sc[5];
    c.lp[0] = 1.0;
    for (int l = 1; l < LANES; ++l) c.lp[l] = c.lp[l - 1] * c.pw[E];
}

// T = tridiag(-beta, D, -beta) = kappa (I - rho S)(I - rho S^T) + kappa rho^2 e0 e0^T
void build_cset_heat1d(CSet &c, std::vector<double> &tab, int n, double fac, double dt) {
    const double beta = dt * fac;
    const double D = dt * (2.0 * fac) + 1.0;
    const double s = std::sqrt((D - 2.0 * beta) * (D + 2.0 * beta));
    const double kappa = 0.5 * (D + s);
    const double rho = beta / kappa;
    c.ik = 1.0 / kappa;
    c.scal = 0.0;
    cset_powers(c, rho);
    tab.assign(n, 0.0);
    std::vector<double> y(n);
    y[0] = 1.0;
    for (int j = 1; j < n; ++j) y[j] = rho * y[j - 1];
    double z = y[n - 1];
    tab[n - 1] = z;
    for (int j = n - 2; j >= 0; --j) {
        z = std::fma(rho, z, y[j]);
        tab[j] = z;
    }
    const double kr2 = beta * rho;
    const double w0 = tab[0] * c.ik;
    const double gamma = kr2 / (1.0 + kr2 * w0);
    for (int j = 0; j < n; ++j) tab[j] = gamma * (tab[j] * c.ik);
}

// (1+alpha) x_j - alpha x_{j-1 mod n} = u_j ; r = alpha/D ; x_j = y_j + r^(j+1) x_{n-1}, x_{n-1} = y_{n-1}/(1 - r^n)
void build_cset_advection1d(CSet &c, std::vector<double> &tab, int n, double fac, double dt) {
    const double alpha = dt * fac;
    const double D = alpha + 1.0;
    const double r = alpha / D;
    c.ik = 1.0 / D;
    cset_powers(c, r);
    tab.assign(n, 0.0);
    double p = r;
    for (int j = 0; j < n; ++j) {
        tab[j] = p;
        p = p * r;
    }
    c.scal = 1.0 / (1.0 - tab[n - 1]);
}

template <typename T>
int dev_upload(Level &lv, hipStream_t st, const std::vector<T> &h, T **out) {
    void *d = nullptr;
    const size_t bytes = sizeof(T) * (h.empty() ? 1 : h.size());
    HIP_TRY(hipMalloc(&d, bytes));
    lv.allocs.push_back(d);
    if (!h.empty()) {
        HIP_TRY(hipMemcpyAsync(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    *out = static_cast<T *>(d);
    return 0;
}

size_t smem_bytes(int G) { return (size_t)G * WAVE_TILE_BYTES + (2 * MAX_G + 2) * sizeof(double); }

template <typename K>
int allow_big_lds(K kernel) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem_bytes(MAX_G)));
    return 0;
}

bool g_attr_done = false;
int setup_kernel_attrs() {
    if (g_attr_done) return 0;
    int rc;
    if ((rc = allow_big_lds(relax_kernel<MGRIT_HIP_STEPPER_HEAT1D>))) return rc;
    if ((rc = allow_big_lds(relax_kernel<MGRIT_HIP_STEPPER_ADVECTION1D>))) return rc;
    if ((rc = allow_big_lds(residual_kernel<MGRIT_HIP_STEPPER_HEAT1D>))) return rc;
    if ((rc = allow_big_lds(residual_kernel<MGRIT_HIP_STEPPER_ADVECTION1D>))) return rc;
    if ((rc = allow_big_lds(fas_fine_kernel<MGRIT_HIP_STEPPER_HEAT1D>))) return rc;
    if ((rc = allow_big_lds(fas_fine_kernel<MGRIT_HIP_STEPPER_ADVECTION1D>))) return rc;
    if ((rc = allow_big_lds(fas_coarse_kernel<MGRIT_HIP_STEPPER_HEAT1D>))) return rc;
    if ((rc = allow_big_lds(fas_coarse_kernel<MGRIT_HIP_STEPPER_ADVECTION1D>))) return rc;
    if ((rc = allow_big_lds(jump_kernel))) return rc;
    g_attr_done = true;
    return 0;
}

int check_level(mgrit_hip_engine *e, int lvl, bool need_set = true) {
    if (!e) return fail(MGRIT_HIP_EINVAL, "null engine");
    if (lvl < 0 || lvl >= e->n_levels) return fail(MGRIT_HIP_EINVAL, "level %d out of range [0,%d)", lvl, e->n_levels);
    if (need_set && !e->L[lvl].set) return fail(MGRIT_HIP_EINVAL, "level %d has no stepper", lvl);
    return 0;
}

int level_common(mgrit_hip_engine *e, int lvl, int kind, int n_pts, const double *t_local, int n, int ld, double fac,
                 int K, const double *s, const double *tau) {
    int rc = check_level(e, lvl, false);
    if (rc) return rc;
    if (n < 1 || n > MGRIT_HIP_MAX_N)
        return fail(MGRIT_HIP_EUNSUPPORTED, "n=%d DOFs per time point outside [1,%d] (register-resident stepper)", n,
                    MGRIT_HIP_MAX_N);
    if (ld < n || (ld % 16) != 0) return fail(MGRIT_HIP_EINVAL, "ld=%d must be a multiple of 16 and >= n=%d", ld, n);
    if (n_pts < 0 || (n_pts > 0 && !t_local)) return fail(MGRIT_HIP_EINVAL, "bad local time grid");
    if (K < 0 || K > 8 || (K > 0 && (!s || !tau))) return fail(MGRIT_HIP_EINVAL, "bad forcing description (K=%d)", K);
    if ((rc = setup_kernel_attrs())) return rc;
    Level &lv = e->L[lvl];
    if (lv.set) return fail(MGRIT_HIP_EINVAL, "level %d already described", lvl);
    const int G = (n + GROUP - 1) / GROUP, T = G * LANES;
    lv.G = G;
    LevelDev &d = lv.dev;
    d.n = n; d.ld = ld; d.T = T; d.n_pts = n_pts; d.K = K; d.kind = kind;
    // coefficient sets keyed by the bit pattern of dt = t[i] - t[i-1] (the reference uses each step's own dt)
    std::vector<double> dts(n_pts > 0 ? n_pts : 0, 0.0), uniq;
    std::vector<int32_t> cidx(n_pts > 0 ? n_pts : 0, 0);
    for (int i = 1; i < n_pts; ++i) {
        const double dt = t_local[i] - t_local[i - 1];
        dts[i] = dt;
        int found = -1;
        for (size_t q = 0; q < uniq.size(); ++q)
            if (std::memcmp(&uniq[q], &dt, sizeof(double)) == 0) { found = (int)q; break; }
        if (found < 0) {
            if (uniq.size() >= 4096) return fail(MGRIT_HIP_EUNSUPPORTED, "more than 4096 distinct time-step sizes on level %d", lvl);
            found = (int)uniq.size();
            uniq.push_back(dt);
        }
        cidx[i] = found;
    }
    if (n_pts > 0) cidx[0] = 0;
    lv.n_csets = (int)uniq.size();
    std::vector<CSet> cs(uniq.size());
    std::vector<double> tabT(uniq.size() * (size_t)E * T, 0.0), tab;
    for (size_t q = 0; q < uniq.size(); ++q) {
        std::memset(&cs[q], 0, sizeof(CSet));
        if (kind == MGRIT_HIP_STEPPER_HEAT1D) build_cset_heat1d(cs[q], tab, n, fac, uniq[q]);
        else build_cset_advection1d(cs[q], tab, n, fac, uniq[q]);
        for (int j = 0; j < n; ++j) tabT[q * (size_t)E * T + (size_t)(j % E) * T + (j / E)] = tab[j];
    }
    std::vector<double> sT((size_t)(K > 0 ? K : 0) * E * T, 0.0), tauv;
    for (int kk = 0; kk < K; ++kk)
        for (int j = 0; j < n; ++j) sT[((size_t)kk * E + (j % E)) * T + (j / E)] = s[(size_t)kk * n + j];
    if (K > 0) tauv.assign(tau, tau + (size_t)K * n_pts);
    int32_t *d_cidx; double *d_dt, *d_tau, *d_sT, *d_tabT; CSet *d_cs;
    if ((rc = dev_upload(lv, e->stream, cidx, &d_cidx))) return rc;
    if ((rc = dev_upload(lv, e->stream, dts, &d_dt))) return rc;
    if ((rc = dev_upload(lv, e->stream, tauv, &d_tau))) return rc;
    if ((rc = dev_upload(lv, e->stream, sT, &d_sT))) return rc;
    if ((rc = dev_upload(lv, e->stream, cs, &d_cs))) return rc;
    if ((rc = dev_upload(lv, e->stream, tabT, &d_tabT))) return rc;
    d.cidx = d_cidx; d.dt = d_dt; d.tau = d_tau; d.sT = d_sT; d.cs = d_cs; d.tabT = d_tabT;
    lv.set = true;
    return 0;
}

int get_runs(mgrit_hip_engine *e, int lvl, int id, RunList **out) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (id < 0 || id >= (int)e->L[lvl].runs.size()) return fail(MGRIT_HIP_EINVAL, "bad run-list id %d on level %d", id, lvl);
    *out = &e->L[lvl].runs[id];
    return 0;
}

int get_pairs(mgrit_hip_engine *e, int lvl, int id, PairList **out) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (lvl + 1 >= e->n_levels || !e->L[lvl + 1].set) return fail(MGRIT_HIP_EINVAL, "level %d has no coarser level", lvl);
    if (id < 0 || id >= (int)e->L[lvl].pairs.size()) return fail(MGRIT_HIP_EINVAL, "bad pair-list id %d on level %d", id, lvl);
    *out = &e->L[lvl].pairs[id];
    return 0;
}

int check_bound(const Level &lv, bool need_vg) {
    if (!lv.dev.u) return fail(MGRIT_HIP_EINVAL, "state slabs not bound");
    if (need_vg && (!lv.dev.v || !lv.dev.g)) return fail(MGRIT_HIP_EINVAL, "v/g slabs not bound");
    return 0;
}

#define LAUNCH_BY_KIND(kernel, lv, grid, ...)                                                                     \
    do {                                                                                                          \
        if ((lv).dev.kind == MGRIT_HIP_STEPPER_HEAT1D)                                                             \
            hipLaunchKernelGGL(kernel<MGRIT_HIP_STEPPER_HEAT1D>, dim3(grid), dim3((lv).dev.T), smem_bytes((lv).G), \
                               e->stream, __VA_ARGS__);                                                           \
        else                                                                                                      \
            hipLaunchKernelGGL(kernel<MGRIT_HIP_STEPPER_ADVECTION1D>, dim3(grid), dim3((lv).dev.T),                \
                               smem_bytes((lv).G), e->stream, __VA_ARGS__);                                       \
        HIP_TRY(hipGetLastError());                                                                               \
    } while (0)

}  // namespace

// ===============================================================================================================
// C ABI
// ===============================================================================================================
extern "C" {

int mgrit_hip_abi_version(void) { return MGRIT_HIP_ABI_VERSION; }
const char *mgrit_hip_last_error(void) { return g_err.c_str(); }

int mgrit_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mgrit_hip_create(mgrit_hip_engine **out, int n_levels, void *stream) {
    if (!out || n_levels < 1 || n_levels > 64) return fail(MGRIT_HIP_EINVAL, "bad arguments to mgrit_hip_create");
    if (mgrit_hip_device_count() < 1) return fail(MGRIT_HIP_ENODEV, "no HIP device visible: the MI355X engine has no CPU fallback");
    mgrit_hip_engine *e = new mgrit_hip_engine();
    e->n_levels = n_levels;
    e->stream = static_cast<hipStream_t>(stream);
    e->L.resize(n_levels);
    *out = e;
    return 0;
}

int mgrit_hip_destroy(mgrit_hip_engine *e) {
    if (!e) return 0;
    (void)hipStreamSynchronize(e->stream);
    for (auto &lv : e->L) {
        for (void *p : lv.allocs) (void)hipFree(p);
        if (lv.scratch) (void)hipFree(lv.scratch);
    }
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    delete e;
    return 0;
}

int mgrit_hip_sync(mgrit_hip_engine *e) {
    if (!e) return fail(MGRIT_HIP_EINVAL, "null engine");
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

int mgrit_hip_level_heat1d(mgrit_hip_engine *e, int lvl, int n_pts_local, const double *t_local, int n, int ld,
                           double fac, int K, const double *s, const double *tau) {
    return level_common(e, lvl, MGRIT_HIP_STEPPER_HEAT1D, n_pts_local, t_local, n, ld, fac, K, s, tau);
}

int mgrit_hip_level_advection1d(mgrit_hip_engine *e, int lvl, int n_pts_local, const double *t_local, int n, int ld,
                                double fac) {
    return level_common(e, lvl, MGRIT_HIP_STEPPER_ADVECTION1D, n_pts_local, t_local, n, ld, fac, 0, nullptr, nullptr);
}

int mgrit_hip_level_bind(mgrit_hip_engine *e, int lvl, double *u, double *v, double *g) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (!u && e->L[lvl].dev.n_pts > 0) return fail(MGRIT_HIP_EINVAL, "u slab is null");
    e->L[lvl].dev.u = u; e->L[lvl].dev.v = v; e->L[lvl].dev.g = g;
    return 0;
}

int mgrit_hip_level_transfer(mgrit_hip_engine *e, int lvl, int kind) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (lvl + 1 >= e->n_levels || !e->L[lvl + 1].set) return fail(MGRIT_HIP_EINVAL, "describe level %d before its transfer", lvl + 1);
    const int nf = e->L[lvl].dev.n, nc = e->L[lvl + 1].dev.n;
    if (kind == MGRIT_HIP_TRANSFER_COPY) {
        if (nf != nc) return fail(MGRIT_HIP_EINVAL, "copy transfer needs equal DOFs (%d vs %d)", nf, nc);
    } else if (kind == MGRIT_HIP_TRANSFER_HEAT1D) {
        if (nf != 2 * nc + 1) return fail(MGRIT_HIP_EINVAL, "full-weighting transfer needs n_fine = 2*n_coarse+1 (%d vs %d)", nf, nc);
    } else return fail(MGRIT_HIP_EINVAL, "unknown transfer kind %d", kind);
    e->L[lvl].transfer = kind;
    return 0;
}

int mgrit_hip_runs_create(mgrit_hip_engine *e, int lvl, int n_runs, const int32_t *start, const int32_t *len, int *id_out) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (n_runs < 0 || !id_out || (n_runs > 0 && (!start || !len))) return fail(MGRIT_HIP_EINVAL, "bad run list");
    Level &lv = e->L[lvl];
    for (int r = 0; r < n_runs; ++r)
        if (start[r] < 1 || len[r] < 1 || start[r] + len[r] > lv.dev.n_pts)
            return fail(MGRIT_HIP_EINVAL, "run %d = [%d,+%d) outside the local grid of %d points (predecessor required)", r,
                        start[r], len[r], lv.dev.n_pts);
    RunList rl;
    rl.n = n_runs;
    std::vector<int32_t> hs(start, start + n_runs), hl(len, len + n_runs);
    if ((rc = dev_upload(lv, e->stream, hs, &rl.d_start))) return rc;
    if ((rc = dev_upload(lv, e->stream, hl, &rl.d_len))) return rc;
    lv.runs.push_back(rl);
    *id_out = (int)lv.runs.size() - 1;
    return 0;
}

int mgrit_hip_pairs_create(mgrit_hip_engine *e, int lvl, int n_pairs, const int32_t *fine_idx, const int32_t *coarse_idx,
                           int *id_out) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (lvl + 1 >= e->n_levels || !e->L[lvl + 1].set) return fail(MGRIT_HIP_EINVAL, "level %d has no described coarser level", lvl);
    if (n_pairs < 0 || !id_out || (n_pairs > 0 && (!fine_idx || !coarse_idx))) return fail(MGRIT_HIP_EINVAL, "bad pair list");
    Level &lv = e->L[lvl];
    const Level &lc = e->L[lvl + 1];
    for (int p = 0; p < n_pairs; ++p)
        if (fine_idx[p] < 0 || fine_idx[p] >= lv.dev.n_pts || coarse_idx[p] < 0 || coarse_idx[p] >= lc.dev.n_pts)
            return fail(MGRIT_HIP_EINVAL, "pair %d = (%d,%d) outside the local grids (%d,%d)", p, fine_idx[p], coarse_idx[p],
                        lv.dev.n_pts, lc.dev.n_pts);
    PairList pl;
    pl.n = n_pairs;
    std::vector<int32_t> hf(fine_idx, fine_idx + n_pairs), hc(coarse_idx, coarse_idx + n_pairs), iota(n_pairs);
    for (int p = 0; p < n_pairs; ++p) iota[p] = p;
    if ((rc = dev_upload(lv, e->stream, hf, &pl.d_fine))) return rc;
    if ((rc = dev_upload(lv, e->stream, hc, &pl.d_coarse))) return rc;
    if ((rc = dev_upload(lv, e->stream, iota, &pl.d_iota))) return rc;
    lv.pairs.push_back(pl);
    *id_out = (int)lv.pairs.size() - 1;
    return 0;
}

int mgrit_hip_relax(mgrit_hip_engine *e, int lvl, int runs_id, int mode, double weight_c) {
    RunList *rl;
    int rc = get_runs(e, lvl, runs_id, &rl);
    if (rc) return rc;
    Level &lv = e->L[lvl];
    if (mode != MGRIT_HIP_RELAX_F && mode != MGRIT_HIP_RELAX_C) return fail(MGRIT_HIP_EINVAL, "bad relax mode %d", mode);
    if ((rc = check_bound(lv, lvl > 0))) return rc;
    if (rl->n == 0) return 0;
    if (e->timing) {
        if (!e->ev0) { HIP_TRY(hipEventCreate(&e->ev0)); HIP_TRY(hipEventCreate(&e->ev1)); }
        HIP_TRY(hipEventRecord(e->ev0, e->stream));
    }
    LAUNCH_BY_KIND(relax_kernel, lv, rl->n, lv.dev, rl->d_start, rl->d_len, lvl > 0 ? 1 : 0, mode, weight_c, 1.0 - weight_c);
    if (e->timing) { HIP_TRY(hipEventRecord(e->ev1, e->stream)); e->ev_valid = true; }
    return 0;
}

int mgrit_hip_residual(mgrit_hip_engine *e, int lvl, int runs_id, double *sumsq_out) {
    RunList *rl;
    int rc = get_runs(e, lvl, runs_id, &rl);
    if (rc) return rc;
    Level &lv = e->L[lvl];
    if ((rc = check_bound(lv, false))) return rc;
    if (rl->n == 0) return 0;
    if (!sumsq_out) return fail(MGRIT_HIP_EINVAL, "null output");
    LAUNCH_BY_KIND(residual_kernel, lv, rl->n, lv.dev, rl->d_start, sumsq_out);
    return 0;
}

int mgrit_hip_jump(mgrit_hip_engine *e, int lvl, int runs_id, const double *prev, double *sumsq_out) {
    RunList *rl;
    int rc = get_runs(e, lvl, runs_id, &rl);
    if (rc) return rc;
    Level &lv = e->L[lvl];
    if ((rc = check_bound(lv, false))) return rc;
    if (rl->n == 0) return 0;
    if (!sumsq_out || !prev) return fail(MGRIT_HIP_EINVAL, "null argument");
    hipLaunchKernelGGL(jump_kernel, dim3(rl->n), dim3(lv.dev.T), smem_bytes(lv.G), e->stream, lv.dev, rl->d_start, prev, sumsq_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mgrit_hip_restrict_u(mgrit_hip_engine *e, int lvl, int pairs_id) {
    PairList *pl;
    int rc = get_pairs(e, lvl, pairs_id, &pl);
    if (rc) return rc;
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    if ((rc = check_bound(lf, false)) || (rc = check_bound(lc, false))) return rc;
    if (pl->n == 0) return 0;
    dim3 grid(pl->n, (lc.dev.n + 255) / 256);
    hipLaunchKernelGGL(restrict_rows_kernel, grid, dim3(256), 0, e->stream, lf.dev.u, lf.dev.ld, pl->d_fine, lc.dev.u, lc.dev.ld,
                       pl->d_coarse, lc.dev.n, lf.transfer);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mgrit_hip_copy_u_to_v(mgrit_hip_engine *e, int lvl_coarse) {
    int rc = check_level(e, lvl_coarse);
    if (rc) return rc;
    Level &lc = e->L[lvl_coarse];
    if ((rc = check_bound(lc, true))) return rc;
    if (lc.dev.n_pts == 0) return 0;
    HIP_TRY(hipMemcpyAsync(lc.dev.v, lc.dev.u, sizeof(double) * (size_t)lc.dev.n_pts * lc.dev.ld, hipMemcpyDeviceToDevice, e->stream));
    return 0;
}

int mgrit_hip_fas_rhs(mgrit_hip_engine *e, int lvl, int pairs_id) {
    PairList *pl;
    int rc = get_pairs(e, lvl, pairs_id, &pl);
    if (rc) return rc;
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    if ((rc = check_bound(lf, lvl > 0)) || (rc = check_bound(lc, true))) return rc;
    if (pl->n == 0) return 0;
    if (lf.transfer == MGRIT_HIP_TRANSFER_COPY) {
        LAUNCH_BY_KIND(fas_fine_kernel, lf, pl->n, lf.dev, pl->d_fine, pl->d_coarse, lc.dev.g, lc.dev.ld, lvl > 0 ? 1 : 0);
    } else {
        if (lf.scratch_rows < (size_t)pl->n) {
            if (lf.scratch) HIP_TRY(hipFree(lf.scratch));
            lf.scratch = nullptr;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&lf.scratch), sizeof(double) * (size_t)pl->n * lf.dev.ld));
            lf.scratch_rows = pl->n;
        }
        LAUNCH_BY_KIND(fas_fine_kernel, lf, pl->n, lf.dev, pl->d_fine, pl->d_iota, lf.scratch, lf.dev.ld, lvl > 0 ? 1 : 0);
        dim3 grid(pl->n, (lc.dev.n + 255) / 256);
        hipLaunchKernelGGL(restrict_rows_kernel, grid, dim3(256), 0, e->stream, lf.scratch, lf.dev.ld, pl->d_iota, lc.dev.g,
                           lc.dev.ld, pl->d_coarse, lc.dev.n, lf.transfer);
        HIP_TRY(hipGetLastError());
    }
    LAUNCH_BY_KIND(fas_coarse_kernel, lc, pl->n, lc.dev, pl->d_coarse);
    return 0;
}

static int interp_common(mgrit_hip_engine *e, int lvl, int pairs_id, int mode) {
    PairList *pl;
    int rc = get_pairs(e, lvl, pairs_id, &pl);
    if (rc) return rc;
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    if ((rc = check_bound(lf, false)) || (rc = check_bound(lc, mode == 1))) return rc;
    if (pl->n == 0) return 0;
    dim3 grid(pl->n, (lf.dev.n + 255) / 256);
    hipLaunchKernelGGL(interp_rows_kernel, grid, dim3(256), 0, e->stream, lf.dev.u, lf.dev.ld, pl->d_fine, lc.dev.u, lc.dev.v,
                       lc.dev.ld, pl->d_coarse, lf.dev.n, lc.dev.n, lf.transfer, mode);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mgrit_hip_error_correction(mgrit_hip_engine *e, int lvl, int pairs_id) { return interp_common(e, lvl, pairs_id, 1); }
int mgrit_hip_interpolate(mgrit_hip_engine *e, int lvl, int pairs_id) { return interp_common(e, lvl, pairs_id, 0); }

int mgrit_hip_set_timing(mgrit_hip_engine *e, int enabled) {
    if (!e) return fail(MGRIT_HIP_EINVAL, "null engine");
    e->timing = enabled != 0;
    e->ev_valid = false;
    return 0;
}

int mgrit_hip_last_kernel_ms(mgrit_hip_engine *e, float *ms) {
    if (!e || !ms) return fail(MGRIT_HIP_EINVAL, "null argument");
    if (!e->ev_valid) return fail(MGRIT_HIP_EINVAL, "no timed launch recorded");
    HIP_TRY(hipEventSynchronize(e->ev1));
    HIP_TRY(hipEventElapsedTime(ms, e->ev0, e->ev1));
    return 0;
}

}  // extern "C"
