// mgrit_hip.hip -- MI355X (gfx950 / CDNA4) MGRIT relaxation engine: HIP kernels + the C ABI of include/mgrit_hip.h.
//
// Replaces the per-time-point Python loops of the reference's hot path (src/pymgrit/core/mgrit.py:292-549,715-726)
// and the per-step SuperLU solves of heat/heat_1d.py:198-217 and advection/advection_1d.py:129-143.
//
// Data layout: every level keeps its time-point states in one float64 slab [n_local_points][ld] in HBM; inside a row
// the x-values are stored in "lane-blocked" order (row_pos below) so that the lane owning 16 consecutive x-values
// fetches them with 8 fully coalesced 16-byte loads -- no LDS transpose, no strided access.
// Execution model: ONE workgroup per run of consecutive time points (an F-interval, a C-point, or the coarsest-level
// chain). The workgroup keeps the whole state vector in registers (16 consecutive x per lane, 64 lanes per wave,
// up to 16 waves = 16384 x), the rank-one correction table of the current time-step size in LDS, and applies Phi as
// two constant-coefficient first-order recurrences (forward, backward) + a rank-one correction -- each recurrence is
// a chunked scan: lane-local FMA chain, row-wise Kogge-Stone over DPP row shifts + two readlane row broadcasts inside
// a wave (no LDS), serial carry across waves through LDS (one barrier).
// The arithmetic (operation order, FMA placement, reduction trees) is specified in DESIGN.md section 3 and must
// match oracle/mgrit_oracle.c variant 1 bit for bit: compile with -ffp-contract=off; every FMA is explicit.
//
// gfx950 only. No CUDA paths, no fallbacks: every entry point fails when no HIP device is usable.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is loaded on first use (mgrit_hip_comm.inc)
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mgrit_hip.h"

namespace {

constexpr int E = MGRIT_HIP_E;          // elements per lane
constexpr int LANES = 64;               // lanes per wave
constexpr int GROUP = E * LANES;        // elements per wave
constexpr int MAX_G = MGRIT_HIP_MAX_N / GROUP;  // 16 waves

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
    double pi_full, pi_last;  // heat: Pt[0] of a full / of the last group; advection: pi_last = rho^(((n-1) mod 16) + 1)
    double pg[MAX_G], qg[MAX_G], qg2[MAX_G];  // heat: per-group factors of the rank-one correction (build_cset_heat1d)
    double pw[E + 1];   // rho^k
    double sc[6];       // rho^(E*2^s)
    double lp[LANES];   // rho^(E*l)
};

// Row storage order ("lane-blocked"): wave w of the workgroup owns the natural indices
// [1024w, 1024w+1024); inside that block lane l owns 16 consecutive values j = 1024w + 16l + 2q + r (q = 0..7, r = 0..1)
// and the pair (q) of lane l lives at 16-byte slot (8w + q)*64 + l of the row. One wave instruction therefore moves one
// contiguous 1 KiB segment (64 lanes x 16 B), the 8 segments of a wave are 1 KiB apart (immediate offsets), and no LDS
// transpose or strided access is needed to give every lane its 16 consecutive x-values. Row stride ld = 1024*ceil(n/1024).
struct LevelDev {
    double *u, *v, *g;
    const int32_t *cidx;  // [n_pts] coefficient set of the step (i-1 -> i)
    const double *dt;     // [n_pts]
    const double *tc;     // [K][n_pts] tau_k(t_i) * dt_i
    const double2 *sP;    // [K][ld/2] forcing space factors, row storage order
    const CSet *cs;       // [n_csets]
    const double2 *tabP;  // [n_csets][ld/2] rank-one correction table, row storage order
    const double2 *ptP;   // [n_csets][2][512] heat: group-local backward scan of rho^(j'+1) (full group, last group)
    const int32_t *cidx2; // two-point steppers: [n_pts][2] coefficient sets of the two half-solves (then tc is [K][n_pts][2])
    const double *hc;     // two-point BDF2: [n_pts][4] = (a, nb) of the first and of the second half-solve
    const double *fb;     // general (non-separable) forcing, FORCE == 3: [n_pts][ld] rows rhs(x, t_i)*dt_i in row storage
                          // order (the product heat_1d.py:213 adds to u_start), caller-owned like the state slabs; else null
    const double *chT;    // overlapped chain, null when the level does not qualify: V3 row [ld], then
                          // Q1, V1, V2 as [2][1024] each (full / last group), then a1 b1 a2 b2 [2] each, a3[16], b3[16]
    int n, ld, T, n_pts, K, kind;
    int one_cset;         // every step of the level has the same size: coefficient set 0, cidx is not read
    int stream_rows;      // the level's slabs are far larger than the caches (n_pts * ld * 8 B > 256 MB): rows written by the
                          // whole-level passes are stored with the non-temporal hint (store_row_nt)
    // launch-time fields (set per launch by the host, not part of the level description):
    int *sched;           // null: persistent workgroups walk their items with stride gridDim.x. Else {next, xcc0, done}: items
                          // are drawn from a device-wide queue head (see WgQueue)
    int xcc0_limit;       // with sched: how many workgroups may stay on XCD 0 (the others there leave at once, so that the
                          // chain workers of a planned cycle find free CUs on that XCD); < 0: no limit
};

__host__ __device__ __forceinline__ int row_pos(int j) {
    return ((((j >> 10) * 8 + ((j & 15) >> 1)) * 64 + ((j >> 4) & 63)) << 1) + (j & 1);
}
__host__ __device__ __forceinline__ int row_nat(int p) {
    const int slot = p >> 1, lane = slot & 63, q = (slot >> 6) & 7, w = slot >> 9;
    return 1024 * w + 16 * lane + 2 * q + (p & 1);
}
// 16-byte slot of pair q for thread t = 64*wave + lane
__device__ __forceinline__ unsigned slot0(int t) { return (unsigned)(((t >> 6) << 9) + (t & 63)); }

// ---------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------
// LDS-only workgroup barrier. __syncthreads() also waits for vmcnt(0) (its fence covers global memory), which would drain
// the GEMM's register prefetch of the next K step in front of every barrier. Only the GEMM uses it: in the 1-D steppers it
// bought nothing measurable (their loads are consumed before the next barrier anyway).
// (Builtins, not inline asm: an `asm volatile("s_waitcnt lgkmcnt(0); s_barrier")` in this place was observed to let waves
// read the LDS group totals of a step before they were written in one kernel -- the compiler does not treat an asm
// string as a barrier when it schedules LDS accesses.)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0), vmcnt / expcnt untouched (gfx9 encoding)
    __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ int xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return (int)(x & 0xfu);
}

// One device-scope add by ONE lane of the calling wave, without a divergent branch: every lane of the wave issues the
// atomic, lane 0 adds `inc` to *word, lanes 1..63 add 0 to a word of their own in the 256-byte dummy row behind the counters
// (one fully coalesced atomic wave-instruction). Returns the value lane 0 got back, wave-uniform. Call it from code that the
// whole wave executes (branch on wave-uniform scalars only). Why not `if (t == 0) atomicAdd(...)`: with that branch at the top
// of the item loop of the 128-VGPR sweep kernels the compiler (ROCm 7.2) placed a live-range copy of a loop-invariant VGPR in
// front of the `s_or_b64 exec` that re-joins the branch, i.e. executed it for lane 0 only -- every other lane lost the value.
constexpr int SCHED_DUMMY = 16;   // ints from the counter block to the dummy row (64 ints)
__device__ __forceinline__ int lane0_add(int *base, int word, int inc, int lane) {
    int *p = lane == 0 ? base + word : base + SCHED_DUMMY + lane;
    const int old = __hip_atomic_fetch_add(p, lane == 0 ? inc : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane(old);
}
// issue only (the returned per-lane value is made uniform later, after the loads that were issued behind it)
__device__ __forceinline__ int lane0_add_issue(int *base, int word, int inc, int lane) {
    int *p = lane == 0 ? base + word : base + SCHED_DUMMY + lane;
    return __hip_atomic_fetch_add(p, lane == 0 ? inc : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Item scheduling of the persistent sweep kernels. Program order (sched == null): workgroup b takes items b, b + gridDim.x, ...
// Planned cycle (sched != null): the sweeps share the chip with the coarsest-level chain, whose 16 single-group workers want
// 16 CUs of ONE XCD (their exchange stays inside that XCD's L2). The dispatcher hands workgroups to XCDs round-robin, so a
// chip-filling sweep would leave no room anywhere: instead, of the sweep workgroups that find themselves on XCD 0 (hardware
// id) only `xcc0_limit` stay, the others return at once, and -- the number of active workgroups no longer being known in
// advance -- every workgroup draws its items from a queue head with one device-scope atomic add per item, issued ahead of
// the item's row loads so that its latency is hidden behind them. Which workgroup processes which item has no effect on the
// results (items are independent). The last workgroup to leave resets the three counters for the next launch on the stream.
// sched = {next, xcc0, done, -, (chain: tickets, done), ..., dummy row at SCHED_DUMMY}
struct WgQueue {
    int *ctr;       // counter block or null
    int cur, par, pending;
    int *slot;      // LDS: two ints
    bool w0;        // this wave is wave 0 of the workgroup (wave-uniform)
    __device__ __forceinline__ void begin(int *ctr_, int xcc0_limit, int *lds2, int t) {
        ctr = ctr_; slot = lds2; par = 0; pending = 0;
        w0 = __builtin_amdgcn_readfirstlane(t >> 6) == 0;
        if (!ctr) { cur = (int)blockIdx.x; return; }
        if (w0) {
            const int lane = t & 63;
            bool stay = true;
            if (xcc0_limit >= 0 && xcc_id() == 0) stay = lane0_add(ctr, 1, 1, lane) < xcc0_limit;
            int first = 0x7fffffff;
            if (stay) first = lane0_add(ctr, 0, 1, lane);
            slot[0] = first;      // every lane of wave 0 stores the same value
        }
        __syncthreads();
        cur = slot[0];
    }
    // call at the top of an item, BEFORE its loads: draws the next item, the value is consumed in advance()
    __device__ __forceinline__ void prefetch(int t) {
        if (ctr && w0) pending = lane0_add_issue(ctr, 0, 1, t & 63);
    }
    __device__ __forceinline__ void advance(int t) {
        if (!ctr) { cur += (int)gridDim.x; return; }
        par ^= 1;
        if (w0) slot[par] = __builtin_amdgcn_readfirstlane(pending);
        __syncthreads();
        cur = slot[par];
    }
    __device__ __forceinline__ void end(int t) {
        if (ctr && w0) {
            const int lane = t & 63;
            if (lane0_add(ctr, 2, 1, lane) == (int)gridDim.x - 1) {   // the last workgroup to leave: clear {next, xcc0, done}
                int *p = lane < 3 ? ctr + lane : ctr + SCHED_DUMMY + lane;
                __hip_atomic_store(p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
};

__device__ __forceinline__ void load_row(const double *__restrict__ row, unsigned s0, double (&x)[E]) {
    const double2 *r2 = reinterpret_cast<const double2 *>(row) + s0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const double2 v = r2[q * 64];
        x[2 * q] = v.x;
        x[2 * q + 1] = v.y;
    }
}

__device__ __forceinline__ void store_row(double *__restrict__ row, unsigned s0, const double (&x)[E]) {
    double2 *r2 = reinterpret_cast<double2 *>(row) + s0;
#pragma unroll
    for (int q = 0; q < 8; ++q) r2[q * 64] = make_double2(x[2 * q], x[2 * q + 1]);
}

// The same store with the non-temporal hint, for the rows the whole-level passes write and nobody reads again while they could
// still sit in L2 (a pass writes gigabytes, an XCD's L2 holds 4 MB): written the plain way they push the tables every Phi reads
// -- forcing factors, Pt, coefficient sets -- out of L2, and the Phi that follows a row store pays for it (measured with
// wall-clock stamps inside cfas_kernel: the Phi after the C-point stores takes 8 us where an F-step takes 4.3). Planned cycle of
// config 3: 8.9 -> 8.55 ms. Not for rows the NEXT sweep of a level reads right away from a small level (relax mode FC got slower).
__device__ __forceinline__ void store_row_nt(double *__restrict__ row, unsigned s0, const double (&x)[E], int streaming) {
    if (!streaming) {   // (wave-uniform) a level that fits the caches: the next sweep finds the row there
        store_row(row, s0, x);
        return;
    }
    typedef double dv2 __attribute__((ext_vector_type(2)));
    dv2 *r2 = reinterpret_cast<dv2 *>(row) + s0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        dv2 v;
        v.x = x[2 * q];
        v.y = x[2 * q + 1];
        __builtin_nontemporal_store(v, r2 + q * 64);
    }
}

// ... and the rows such a pass reads once (measured on config 3, program order: cfas 2.77 -> 2.46 ms, ecfr 1.92 -> 1.73, relax
// mode FC 0.65 -> 0.58; the stand-alone F- / C-relaxation keeps the plain loads: 1.5 % slower with the hint)
__device__ __forceinline__ void load_row_nt(const double *__restrict__ row, unsigned s0, double (&x)[E], int streaming) {
    if (!streaming) {
        load_row(row, s0, x);
        return;
    }
    typedef double dv2 __attribute__((ext_vector_type(2)));
    const dv2 *r2 = reinterpret_cast<const dv2 *>(row) + s0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const dv2 v = __builtin_nontemporal_load(r2 + q * 64);
        x[2 * q] = v.x;
        x[2 * q + 1] = v.y;
    }
}

// cross-lane moves of a double inside one wave, DPP / readlane (no LDS)
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // bound_ctrl: lanes without a source lane read 0, so "old" is never observed; passing the source avoids a v_mov
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_ROW_SHL = 0x100, DPP_ROW_SHR = 0x110, DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

__device__ __forceinline__ double read_lane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

struct Smem {
    double2 *tab;   // [8*T] correction table of the current coefficient set (heat: w-gamma, advection: r^(j+1)), row order;
                    // free for the kernel's own use where every Heat1D step is taken with heat_solve<CLOSED> (cfas_kernel)
    double2 *pt;    // [2][512] heat: group-local backward scan of rho^(j'+1): [0] full group, [1] last group
    double *ga;     // [2][MAX_G] forward group totals A_g, double-buffered by step parity
    double *gb;     // [2][MAX_G] backward group totals B_g (advection: slot [p][0] carries y-hat of the last real element)
    double *lp;     // [LANES] rho^(E*l) of the current coefficient set
    double *red;    // [MAX_G] per-wave partial sums of block_sumsq (own slots: persistent kernels reuse the others)
    double *wf;     // [3][MAX_G] heat: per-group factors pg | qg | qg2 of the rank-one correction (CSet, build_cset_heat1d)
    double *lp2, *wf2;   // lp and wf of ANOTHER level's (only) coefficient set, staged once by kernels that also step that level
};

// wave-uniform scalar coefficients of the current coefficient set, forced into SGPRs (readfirstlane)
struct Coef {
    double rho, ik, scal, pi_full, pi_last;
    double pw[E + 1];
    double sc[4];
    double gcp[4];  // gc^(2^s), gc = rho^1024: cross-group scan coefficients
};

// per-lane powers of the current coefficient set used by the wave scans
struct LaneCoef {
    double f_row, f_hi;   // forward:  lp[(l&15)+1], lp[l-31]
    double b_row, b_lo;   // backward: lp[16-(l&15)], lp[32-l]
    double f_in, b_in;    // carry-in: lp[l], lp[63-l]
};

__device__ __forceinline__ LaneCoef lane_coef(const double *lp, int lane) {
    LaneCoef k;
    // lanes that do not take part in a broadcast stage get a zero coefficient: fma(0, s, a) = a (no select needed)
    const int li = lane & 15;
    k.f_row = ((lane >> 4) & 1) ? lp[li + 1] : 0.0;
    k.f_hi = lane >= 32 ? lp[lane - 31] : 0.0;
    k.b_row = ((lane >> 4) & 1) ? 0.0 : lp[16 - li];
    k.b_lo = lane < 32 ? lp[32 - lane] : 0.0;
    k.f_in = lp[lane];
    k.b_in = lp[LANES - 1 - lane];
    return k;
}

__device__ __forceinline__ double to_sgpr(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Wave-uniform loads of the read-only per-step data of a level (coefficient-set index, forcing coefficient) through the
// scalar cache. Written as plain loads they become VECTOR loads (the kernels store to global memory, so the compiler cannot
// prove the arrays unchanged and refuses s_load), and a vector load's result can only be waited for with s_waitcnt vmcnt,
// which retires in order: the load issued at the top of a step then also waits for the row STORES of the step before it --
// one exposed store round trip per Phi. Scalar loads count on lgkmcnt and leave the stores in flight.
template <typename T>
__device__ __forceinline__ const T *uniform_ptr(const T *p) {   // the address in SGPRs even where the compiler holds it in VGPRs
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return reinterpret_cast<const T *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int ld_uniform(const int32_t *p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(uniform_ptr(p)));
    return v;
}
__device__ __forceinline__ double ld_uniform(const double *p) {
    double v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(uniform_ptr(p)));
    return v;
}

// a device-resident pointer that a kernel earlier on the stream may have changed (plain load, then made wave-uniform)
__device__ __forceinline__ double *ld_uniform_ptr(double *const *p) {
    const unsigned long long a = *reinterpret_cast<const volatile unsigned long long *>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return reinterpret_cast<double *>(((unsigned long long)hi << 32) | lo);
}

// Row accesses with BOTH halves of a row addressed as (uniform base in SGPRs) + (the lane's one 32-bit offset) + immediate: the
// second base, 4 KB on, is laundered through readfirstlane so that the compiler does not fold it back into a 64-bit per-lane
// address (a VGPR pair per row that the 128-VGPR passes cannot afford to keep).
__device__ __forceinline__ void row_load2(const double *__restrict__ row, unsigned s0, double (&x)[E], int streaming) {
    typedef double dv2 __attribute__((ext_vector_type(2)));
    const dv2 *lo = reinterpret_cast<const dv2 *>(row) + s0, *hi = reinterpret_cast<const dv2 *>(uniform_ptr(row + 512)) + s0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const dv2 *p = (q < 4 ? lo : hi) + (q & 3) * 64;
        const dv2 v = streaming ? __builtin_nontemporal_load(p) : *p;
        x[2 * q] = v.x;
        x[2 * q + 1] = v.y;
    }
}
__device__ __forceinline__ void row_store2(double *__restrict__ row, unsigned s0, const double (&x)[E], int streaming) {
    typedef double dv2 __attribute__((ext_vector_type(2)));
    dv2 *lo = reinterpret_cast<dv2 *>(row) + s0, *hi = reinterpret_cast<dv2 *>(const_cast<double *>(uniform_ptr(row + 512))) + s0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        dv2 v;
        v.x = x[2 * q];
        v.y = x[2 * q + 1];
        dv2 *p = (q < 4 ? lo : hi) + (q & 3) * 64;
        if (streaming) __builtin_nontemporal_store(v, p);
        else *p = v;
    }
}

__device__ __forceinline__ Smem carve_smem(char *base, int T, bool with_pt = true) {
    // with_pt = false (Advection1D: no backward scan, no Pt): 16 KB less, which at 8 groups is the difference between one and two
    // workgroups per CU (83 KB against 67 of the CU's 160)
    Smem s;
    s.tab = reinterpret_cast<double2 *>(base);
    s.pt = s.tab + (size_t)8 * T;
    double *tail = reinterpret_cast<double *>(s.pt + (with_pt ? 2 * 512 : 0));
    s.ga = tail;
    s.gb = tail + 2 * MAX_G;
    s.lp = tail + 4 * MAX_G;
    s.red = tail + 4 * MAX_G + LANES;
    s.wf = tail + 5 * MAX_G + LANES;
    s.lp2 = tail + 8 * MAX_G + LANES;
    s.wf2 = tail + 8 * MAX_G + 2 * LANES;
    return s;
}

// Group-local forward scan  y_j = rho*y_{j-1} + d_j  with zero carry into the group: lane-local chain,
// row-wise Kogge-Stone over DPP row shifts, two readlane row broadcasts, lane carry-in. No LDS, no barrier.
// Returns the group total (value at the last element), wave-uniform.
__device__ __forceinline__ double scan_fwd(double (&x)[E], const Coef &c, const LaneCoef &lc, int lane) {
#pragma unroll
    for (int k = 1; k < E; ++k) x[k] = fma(c.rho, x[k - 1], x[k]);
    double a = x[E - 1];
    // row shifts deliver 0 to lanes without a source (bound_ctrl), and fma(sc, 0, a) = a: no selects in the stages
    a = fma(c.sc[0], dpp_mov<DPP_ROW_SHR + 1>(a), a);
    a = fma(c.sc[1], dpp_mov<DPP_ROW_SHR + 2>(a), a);
    a = fma(c.sc[2], dpp_mov<DPP_ROW_SHR + 4>(a), a);
    a = fma(c.sc[3], dpp_mov<DPP_ROW_SHR + 8>(a), a);
    {
        // lane 15 -> row 1, lane 47 -> row 3 (row_bcast15 also hands lane 31 to row 2, whose coefficient is 0), then lane 31 -> rows
        // 2 and 3: the wave-scan broadcasts of DPP, two moves each instead of four readlanes, four moves and two selects
        a = fma(lc.f_row, dpp_mov<DPP_ROW_BCAST15>(a), a);
        a = fma(lc.f_hi, dpp_mov<DPP_ROW_BCAST31>(a), a);
    }
    const double prev = dpp_mov<DPP_WAVE_SHR1>(a);  // lane 0 has no source lane: 0
#pragma unroll
    for (int k = 0; k < E; ++k) x[k] = fma(c.pw[k + 1], prev, x[k]);
    return read_lane(a, LANES - 1);
}

// Group-local backward scan  z_j = rho*z_{j+1} + y_j  with zero carry from the right; returns the value at the first element
__device__ __forceinline__ double scan_bwd(double (&x)[E], const Coef &c, const LaneCoef &lc, int lane) {
#pragma unroll
    for (int k = E - 2; k >= 0; --k) x[k] = fma(c.rho, x[k + 1], x[k]);
    double a = x[0];
    a = fma(c.sc[0], dpp_mov<DPP_ROW_SHL + 1>(a), a);
    a = fma(c.sc[1], dpp_mov<DPP_ROW_SHL + 2>(a), a);
    a = fma(c.sc[2], dpp_mov<DPP_ROW_SHL + 4>(a), a);
    a = fma(c.sc[3], dpp_mov<DPP_ROW_SHL + 8>(a), a);
    {
        const double s16 = read_lane(a, 16), s48 = read_lane(a, 48);
        a = fma(lc.b_row, lane < 32 ? s16 : s48, a);
        const double s32 = read_lane(a, 32);
        a = fma(lc.b_lo, s32, a);
    }
    const double next = dpp_mov<DPP_WAVE_SHL1>(a);  // lane 63 has no source lane: 0
#pragma unroll
    for (int k = 0; k < E; ++k) x[k] = fma(c.pw[E - k], next, x[k]);
    return read_lane(a, 0);
}

// Per-workgroup stepper state that survives across the steps of a run.
struct StepCtx {
    double s0[E];   // forcing space factor of this lane (first term), FORCE == 1
    Coef c;         // scalar coefficients of the resident coefficient set
    int cur;        // coefficient set whose tables are resident in LDS / SGPRs (-1: none)
    int parity;     // double-buffer index of the LDS gather slots
};

// FORCE: 0 = no forcing term, 1 = one separable term (space factor held in registers), 2 = K >= 2 terms (streamed),
// 3 = general forcing: one precomputed row rhs(x, t_i)*dt_i per time point, streamed from HBM (+8 B per DOF and Phi)
template <int KIND, int FORCE>
__device__ __forceinline__ void ctx_init(StepCtx &ctx, const LevelDev &L, int t) {
    ctx.cur = -1;
    ctx.parity = 0;
    if (KIND == MGRIT_HIP_STEPPER_HEAT1D && FORCE == 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double2 v = L.sP[slot0(t) + q * 64];
            ctx.s0[2 * q] = v.x;
            ctx.s0[2 * q + 1] = v.y;
        }
    }
}

// SGPR = true: 1024-thread kernels are VGPR-bound, keep the scalars in SGPRs. SGPR = false: the single-wave chain
// workers have VGPRs to spare and no use for SGPR spill traffic.
template <bool SGPR>
__device__ __forceinline__ double uni(double v) { return SGPR ? to_sgpr(v) : v; }

// SGPR = true: the scalars of the set through the SCALAR cache, six s_load of up to 64 B and ONE wait (measured inside
// cfas_kernel with wall-clock stamps: the same values as ~30 wave-uniform vector loads + readfirstlane took 3.3 us per call --
// a vector load's result retires in order behind every row load and store the wave has in flight --, twice per interval).
typedef int sgpr4 __attribute__((ext_vector_type(4)));
typedef int sgpr8 __attribute__((ext_vector_type(8)));
typedef int sgpr16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ double sgpr_pair(int lo, int hi) { return __hiloint2double(hi, lo); }

template <bool SGPR = true>
__device__ __forceinline__ void load_coef(Coef &c, const CSet *g) {
    if (SGPR) {
        static_assert(offsetof(CSet, pi_full) == 32 && offsetof(CSet, pw) == 48 + 3 * MAX_G * 8 && offsetof(CSet, sc) == offsetof(CSet, pw) + (E + 1) * 8 &&
                      E == 16, "load_coef reads CSet by byte offsets");
        sgpr8 a;      // rho ik scal gc
        sgpr4 b;      // pi_full pi_last
        sgpr16 p0, p1;   // pw[0..7], pw[8..15]
        sgpr8 q;      // sc[0..3]
        double p16;
        asm volatile("s_load_dwordx8 %0, %6, 0x0\n\t"
                     "s_load_dwordx4 %1, %6, 0x20\n\t"
                     "s_load_dwordx16 %2, %6, %7\n\t"
                     "s_load_dwordx16 %3, %6, %8\n\t"
                     "s_load_dwordx2 %4, %6, %9\n\t"
                     "s_load_dwordx8 %5, %6, %10\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(a), "=&s"(b), "=&s"(p0), "=&s"(p1), "=&s"(p16), "=&s"(q)
                     : "s"(uniform_ptr(g)), "n"(offsetof(CSet, pw)), "n"(offsetof(CSet, pw) + 64), "n"(offsetof(CSet, pw) + 128),
                       "n"(offsetof(CSet, sc)));
        c.rho = sgpr_pair(a[0], a[1]); c.ik = sgpr_pair(a[2], a[3]); c.scal = sgpr_pair(a[4], a[5]);
        c.gcp[0] = sgpr_pair(a[6], a[7]);
        c.pi_full = sgpr_pair(b[0], b[1]); c.pi_last = sgpr_pair(b[2], b[3]);
#pragma unroll
        for (int k = 0; k < 8; ++k) { c.pw[k] = sgpr_pair(p0[2 * k], p0[2 * k + 1]); c.pw[8 + k] = sgpr_pair(p1[2 * k], p1[2 * k + 1]); }
        c.pw[16] = p16;
#pragma unroll
        for (int k = 0; k < 4; ++k) c.sc[k] = sgpr_pair(q[2 * k], q[2 * k + 1]);
    } else {
        c.rho = g->rho; c.ik = g->ik; c.scal = g->scal;
        c.pi_full = g->pi_full; c.pi_last = g->pi_last;
        c.gcp[0] = g->gc;
#pragma unroll
        for (int k = 0; k <= E; ++k) c.pw[k] = g->pw[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) c.sc[k] = g->sc[k];
    }
#pragma unroll
    for (int k = 1; k < 4; ++k) c.gcp[k] = c.gcp[k - 1] * c.gcp[k - 1];
}

// d <- u + dt*b(x, t_i), forcing folded as fma(s_k, tau_k*dt, .)  (L.tc[k][i] = tau_k(t_i)*dt_i)
template <int FORCE>
__device__ __forceinline__ void add_forcing(double (&x)[E], const StepCtx &ctx, const LevelDev &L, int i, int t, const Smem &sm) {
    if (FORCE == 4) {   // one separable term whose space factor the kernel has staged in LDS (stage_forcing): measured, streaming it
                        // from L2 in every Phi (FORCE 2) costs 2 us per Phi on an idle chip and 4 us when every CU does it
        const double c0 = ld_uniform(L.tc + i);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double2 sv = sm.tab[slot0(t) + q * 64];
            x[2 * q] = fma(sv.x, c0, x[2 * q]);
            x[2 * q + 1] = fma(sv.y, c0, x[2 * q + 1]);
        }
    } else if (FORCE == 1) {
        const double c0 = ld_uniform(L.tc + i);
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = fma(ctx.s0[k], c0, x[k]);
    } else if (FORCE == 2) {
        for (int kk = 0; kk < L.K; ++kk) {
            const double ck = ld_uniform(L.tc + (size_t)kk * L.n_pts + i);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double2 sv = L.sP[(size_t)kk * 8 * L.T + slot0(t) + q * 64];
                x[2 * q] = fma(sv.x, ck, x[2 * q]);
                x[2 * q + 1] = fma(sv.y, ck, x[2 * q + 1]);
            }
        }
    } else if (FORCE == 3) {   // u + rhs(x, t_i)*dt_i: the reference's own operation order (heat_1d.py:213)
        const double2 *b2 = reinterpret_cast<const double2 *>(L.fb + (size_t)i * L.ld) + slot0(t);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double2 bv = b2[q * 64];
            x[2 * q] = x[2 * q] + bv.x;
            x[2 * q + 1] = x[2 * q + 1] + bv.y;
        }
    }
}

// Cross-group carries. Lane g (g < 16) of each row holds the value of group g; inclusive
// Kogge-Stone scans over the row with coefficients gc^(2^s).
__device__ __forceinline__ double cross_fwd(double a, const Coef &c) {
    a = fma(c.gcp[0], dpp_mov<DPP_ROW_SHR + 1>(a), a);
    a = fma(c.gcp[1], dpp_mov<DPP_ROW_SHR + 2>(a), a);
    a = fma(c.gcp[2], dpp_mov<DPP_ROW_SHR + 4>(a), a);
    a = fma(c.gcp[3], dpp_mov<DPP_ROW_SHR + 8>(a), a);
    return a;
}
__device__ __forceinline__ double cross_bwd(double a, const Coef &c) {
    a = fma(c.gcp[0], dpp_mov<DPP_ROW_SHL + 1>(a), a);
    a = fma(c.gcp[1], dpp_mov<DPP_ROW_SHL + 2>(a), a);
    a = fma(c.gcp[2], dpp_mov<DPP_ROW_SHL + 4>(a), a);
    a = fma(c.gcp[3], dpp_mov<DPP_ROW_SHL + 8>(a), a);
    return a;
}

// heat: A, B = totals of group (lane & 15) (0 beyond G). Outputs: cm = C_wave, zin = Zf_{wave+1}, zf0 = Zf_0.
__device__ __forceinline__ void heat_chains(const Coef &c, double A, double B, int G, int wave, int lane, double &cm,
                                            double &zin, double &zf0) {
    const int li = lane & 15;
    const double I = cross_fwd(A, c);
    const double C = dpp_mov<DPP_ROW_SHR + 1>(I);   // exclusive: lane 0 of the row reads 0
    const double Bt = li < G ? fma(C, li == G - 1 ? c.pi_last : c.pi_full, B) : 0.0;
    const double J = cross_bwd(Bt, c);
    const double Jn = dpp_mov<DPP_ROW_SHL + 1>(J);  // Zf_{g+1}: lane 15 of the row reads 0
    const int w = __builtin_amdgcn_readfirstlane(wave);
    cm = read_lane(C, w);
    zin = read_lane(Jn, w);
    zf0 = read_lane(J, 0);
}

// advection: forward carries only. Returns C_wave; c_last = C_{G-1}.
__device__ __forceinline__ double fwd_chain(const Coef &c, double A, int G, int wave, int lane, double &c_last) {
    const double I = cross_fwd(A, c);
    const double C = dpp_mov<DPP_ROW_SHR + 1>(I);
    c_last = read_lane(C, __builtin_amdgcn_readfirstlane(G - 1));
    return read_lane(C, __builtin_amdgcn_readfirstlane(wave));
}

// v if a > b else 0.0, compare and selects back to back on vcc: written as `a > b ? v : 0.0` the compiler hoists the sixteen
// compares of a finishing pass in front of the workgroup barrier and keeps their lane masks in 32 SGPRs (which it then spills)
__device__ __forceinline__ double zero_unless_gt(double v, int a, int b) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    asm volatile("v_cmp_gt_i32 vcc, %2, %3\n\tv_cndmask_b32 %0, 0, %0, vcc\n\tv_cndmask_b32 %1, 0, %1, vcc"
                 : "+v"(lo), "+v"(hi) : "v"(a), "v"(b) : "vcc");
    return __hiloint2double(hi, lo);
}

// final pass of the heat step for one lane: carries + rank-one correction
__device__ __forceinline__ void heat_finish(double (&x)[E], const Coef &c, double cm, double cb, double z0, int q, double2 w,
                                            double2 p) {
    const double t0 = fma(cm, p.x, x[2 * q]), t1 = fma(cm, p.y, x[2 * q + 1]);
    const double z_0 = fma(c.pw[E - 2 * q], cb, t0), z_1 = fma(c.pw[E - 2 * q - 1], cb, t1);
    x[2 * q] = fma(-z0, w.x, z_0 * c.ik);
    x[2 * q + 1] = fma(-z0, w.y, z_1 * c.ik);
}

// x <- (I + dt L)^{-1} x for the coefficient set resident in c / lc / sm: group-local scans,
// one exchange of the group totals through the LDS slots ga/gb (one barrier), carries + rank-one correction. The correction
// entries w-gamma_j of this lane come from the table in LDS (sm.tab) or, CLOSED, from the closed form
// fma(-Q, rho^(15-k), P * rho^k) with the per-thread factors P = pg[wave] * lp[lane], Q = (qg | qg2)[wave] *
// lp[(l_last - lane) mod 64] -- the very bits build_cset_heat1d puts into the table, at two more instructions per element but
// without 8 B per DOF of LDS (the fused level passes use that space; a coarse-level Phi inside a fine-level kernel needs no
// table traffic at all). With CLOSED the padding positions j >= n of the result are NOT zero (the table's entries there are,
// the expression's are not): padding of a Heat1D row holds unspecified finite values, which the next step's zeroing between
// its two scans, the masked norms (block_sumsq) and the host views ignore.
// LDSBAR: the step's one barrier as lds_barrier() -- it orders the LDS exchange of the group totals and nothing else, so it does not
// wait for the wave's global stores (__syncthreads() does: s_waitcnt vmcnt(0)). For passes that leave whole rows of stores in
// flight behind them and go on with Phi that issue no vector load (the block solve's passes): the rows drain under the arithmetic.
template <bool CLOSED = false, bool LDSBAR = false, bool ONE = false>
__device__ __forceinline__ void heat_solve(double (&x)[E], const Coef &c, const LaneCoef &lc, const Smem &sm, double *ga,
                                           double *gb, int n, int t, int lane, int wave, int G) {
    const int j0 = t * E, li = lane & 15;
    const double a = scan_fwd(x, c, lc, lane);
    // padding positions (j >= n) exist in the LAST group only: a wave-uniform branch, so that the other waves do not spend 32
    // mask reloads and 32 selects per Phi on a condition that is never true for them (the compiler computes the sixteen lane masks
    // once per kernel, keeps them in 32 SGPRs, spills them and reads them back lane by lane in every Phi; a Phi is bound by the CU's
    // issue slots). Measured and not kept: compare + selects back to back on vcc here (zero_unless_gt: 64 fewer spilled SGPRs,
    // but 19 more spilled VGPRs in cfas_kernel -- 1.93 ms against 1.75)
    if (__builtin_amdgcn_readfirstlane(wave) == __builtin_amdgcn_readfirstlane(G) - 1) {
#pragma unroll
        for (int k = 0; k < E; ++k)
            if (j0 + k >= n) x[k] = 0.0;
    }
    const double b = scan_bwd(x, c, lc, lane);
    // ONE group, known at compile time (the TB = 64 instances of the sweeps and blk_one_kernel pass ONE = true): the carries into
    // the only group are zero and Zf_0 is its own backward total -- what heat_chains computes from the totals through LDS, a
    // barrier, two cross-group scans and three lane reads (C_0 = 0, Zf_1 = 0, Zf_0 = fma(0, pi, B_0) + 0 = B_0; a zero may come
    // out with the other sign, nothing else). A lone wave's Phi is a chain of dependent instructions: 1.45 us in the general form,
    // 1.05 us with this and the finishing pass below. ONE is a template argument: every other instance compiles exactly the code it had.
    constexpr bool one_group = ONE;
    if (!one_group && lane == 0) { ga[wave] = a; gb[wave] = b; }
    double P = 0.0, Q = 0.0;
    if (CLOSED) {
        const int wv = __builtin_amdgcn_readfirstlane(wave), l_last = ((n - 1) / E) & (LANES - 1);
        P = sm.wf[wv] * lc.f_in;
        Q = (lane <= l_last ? sm.wf[MAX_G + wv] : sm.wf[2 * MAX_G + wv]) * sm.lp[(l_last - lane) & (LANES - 1)];
    }
    double cm, zin, zf0;
    if (one_group) {
        cm = 0.0; zin = 0.0; zf0 = b;
    } else {
        if (LDSBAR) lds_barrier();
        else __syncthreads();
        heat_chains(c, li < G ? ga[li] : 0.0, li < G ? gb[li] : 0.0, G, wave, lane, cm, zin, zf0);
    }
    const double z0 = zf0 * c.ik;
    const double cb = lc.b_in * zin;
    const double2 *pt = sm.pt + (wave == G - 1 ? 512 : 0) + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        double2 w;
        if (CLOSED) {
            w.x = fma(-Q, c.pw[E - 1 - 2 * q], P * c.pw[2 * q]);
            w.y = fma(-Q, c.pw[E - 2 - 2 * q], P * c.pw[2 * q + 1]);
        } else {
            w = sm.tab[slot0(t) + q * 64];
        }
        if (one_group) {   // heat_finish with both carries zero: fma(0, p, x) = x and fma(pw, 0, x) = x (up to the sign of a zero) -- no Pt read
            x[2 * q] = fma(-z0, w.x, x[2 * q] * c.ik);
            x[2 * q + 1] = fma(-z0, w.y, x[2 * q + 1] * c.ik);
        } else {
            heat_finish(x, c, cm, cb, z0, q, w, pt[q * 64]);
        }
    }
}

// x <- Phi(x) for the step (i-1 -> i) of level L, one workgroup holding the whole vector (heat_1d.py:198-217 /
// advection_1d.py:129-143; arithmetic: DESIGN.md section 3). One workgroup barrier per application.
// PART (Heat1D): 0 = the whole step; 1 = only the coefficient set and the forcing term d = u + dt*b; 2 = only the solve, for a
// caller that puts work of its own between the two (cfas_kernel: its row stores)
template <int KIND, int FORCE, bool CLOSED = false, int PART = 0, bool LDSBAR = false, bool ONE = false>
__device__ __forceinline__ void phi_apply(double (&x)[E], StepCtx &ctx, const LevelDev &L, int i, const Smem &sm, int t,
                                          int lane, int wave, int G) {
    const int ci = (PART == 2 || L.one_cset) ? (PART == 2 ? ctx.cur : 0) : ld_uniform(L.cidx + i);
    if (ci != ctx.cur) {  // (re)load this coefficient set: tables + lane powers -> LDS, scalars -> SGPRs; uniform branch
        __syncthreads();
        const CSet *g = L.cs + ci;
        if (KIND == MGRIT_HIP_STEPPER_HEAT1D && CLOSED) {
            if (t < 3 * MAX_G) sm.wf[t] = (&g->pg[0])[t];   // pg | qg | qg2 are consecutive members
        } else {
            const double2 *src = L.tabP + (size_t)ci * 8 * L.T + slot0(t);
#pragma unroll
            for (int q = 0; q < 8; ++q) sm.tab[slot0(t) + q * 64] = src[q * 64];
        }
        if (t < LANES) sm.lp[t] = g->lp[t];
        // (ONE: a single group's finishing pass has no carries to apply and never reads Pt -- 16 KB less to stage per workgroup)
        if (KIND == MGRIT_HIP_STEPPER_HEAT1D && !ONE && t < 2 * 512) sm.pt[t] = L.ptP[(size_t)ci * 1024 + t];
        if (KIND == MGRIT_HIP_STEPPER_HEAT1D && !ONE && L.T < 1024)
            for (int r = t + L.T; r < 1024; r += L.T) sm.pt[r] = L.ptP[(size_t)ci * 1024 + r];
        load_coef(ctx.c, g);
        ctx.cur = ci;
        __syncthreads();
    }
    const Coef &c = ctx.c;
    const LaneCoef lc = lane_coef(sm.lp, lane);
    const int j0 = t * E, par = ctx.parity, li = lane & 15;
    ctx.parity ^= 1;
    double *ga = sm.ga + par * MAX_G, *gb = sm.gb + par * MAX_G;
    if (KIND == MGRIT_HIP_STEPPER_HEAT1D) {
        static_assert(FORCE != 4 || CLOSED, "FORCE 4 keeps the forcing factor where the correction table would be");
        if (PART != 2) add_forcing<FORCE>(x, ctx, L, i, t, sm);
        if (PART == 1) { ctx.parity ^= 1; return; }   // (the solve of PART 2 flips the parity back to this step's)
        heat_solve<CLOSED, LDSBAR, ONE>(x, c, lc, sm, ga, gb, L.n, t, lane, wave, G);
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = x[k] * c.ik;
        const double a = scan_fwd(x, c, lc, lane);
        const int jl = L.n - 1;
        if (lane == 0) ga[wave] = a;
        if (t == jl / E) {
            double y = 0.0;
#pragma unroll
            for (int k = 0; k < E; ++k)
                if (k == jl % E) y = x[k];
            gb[0] = y;
        }
        __syncthreads();
        double c_last = 0.0;
        const double cm = fwd_chain(c, li < G ? ga[li] : 0.0, G, wave, lane, c_last);
        const int ll = (jl % GROUP) / E;
        // (c.pi_last = c.pw[(n-1) % 16 + 1], picked by the host: indexed here, at a position known only at run time, the whole coefficient set
        // would move from registers into scratch memory)
        const double ylast = fma(c.pi_last, sm.lp[ll] * c_last, gb[0]);
        const double xl = ylast * c.scal;
        const double cf = lc.f_in * cm;
        // (padding positions exist in the last group only: a wave-uniform branch, see heat_solve)
        if (__builtin_amdgcn_readfirstlane(wave) == __builtin_amdgcn_readfirstlane(G) - 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double2 w = sm.tab[slot0(t) + q * 64];
                const double y0 = fma(c.pw[2 * q + 1], cf, x[2 * q]), y1 = fma(c.pw[2 * q + 2], cf, x[2 * q + 1]);
                x[2 * q] = (j0 + 2 * q < L.n) ? fma(w.x, xl, y0) : 0.0;
                x[2 * q + 1] = (j0 + 2 * q + 1 < L.n) ? fma(w.y, xl, y1) : 0.0;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double2 w = sm.tab[slot0(t) + q * 64];
                const double y0 = fma(c.pw[2 * q + 1], cf, x[2 * q]), y1 = fma(c.pw[2 * q + 2], cf, x[2 * q + 1]);
                x[2 * q] = fma(w.x, xl, y0);
                x[2 * q + 1] = fma(w.y, xl, y1);
            }
        }
    }
}

// sum of squares of the lane-blocked vector r with the spec's reduction tree; result valid in thread 0
__device__ __forceinline__ double block_sumsq(const double (&r)[E], const Smem &sm, int n, int t, int lane, int wave, int G) {
    double acc = 0.0;
    const int kv = n - t * E;   // padding positions (unspecified values, heat_solve) do not count
    if (__builtin_amdgcn_readfirstlane(wave) == __builtin_amdgcn_readfirstlane(G) - 1) {   // (they exist in the last group only)
#pragma unroll
        for (int k = 0; k < E; ++k) acc = fma(zero_unless_gt(r[k], kv, k), r[k], acc);
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) acc = fma(r[k], r[k], acc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = acc + __shfl_xor(acc, off);
    __syncthreads();
    if (lane == 0) sm.red[wave] = acc;
    __syncthreads();
    double tot = 0.0;
    if (t == 0)
        for (int g = 0; g < G; ++g) tot = tot + sm.red[g];
    return tot;
}

// ---------------------------------------------------------------------------------------------------------------
// kernels (one workgroup per run / pair)
// ---------------------------------------------------------------------------------------------------------------
// FORCE 4: the (one) forcing space factor of level L into the LDS region a closed-form Phi leaves free. Every thread reads
// back only what it wrote itself: no barrier.
template <int FORCE>
__device__ __forceinline__ void stage_forcing(const Smem &sm, const LevelDev &L, unsigned sl) {
    if (FORCE == 4) {
#pragma unroll
        for (int q = 0; q < 8; ++q) sm.tab[sl + q * 64] = L.sP[sl + q * 64];
    }
}

extern __shared__ __attribute__((aligned(16))) char smem_raw[];

#define WG_PROLOGUE_G(WAVE_, G_)                                                           \
    const int t = threadIdx.x, lane = t & 63, wave = (WAVE_), G = (G_);                    \
    const Smem sm = carve_smem(smem_raw, L.T, KIND != MGRIT_HIP_STEPPER_ADVECTION1D);      \
    const unsigned sl = slot0(t);                                                          \
    StepCtx ctx;                                                                           \
    __shared__ int wgq_slot[2];                                                            \
    WgQueue wq;                                                                            \
    (void)wgq_slot; (void)wq;                                                              \
    ctx_init<KIND, FORCE>(ctx, L, t)
#define WG_PROLOGUE WG_PROLOGUE_G(t >> 6, (int)(blockDim.x >> 6))
// an instance compiled for TB threads: TB = 64 is ONE wave (G = 1, wave = 0 as constants); such a kernel also passes ONE = true to
// its Phi, and heat_solve then drops the exchange of the group totals, its barrier and the cross-group scans -- see there
#define WG_PROLOGUE_TB(TB_) WG_PROLOGUE_G((TB_) == LANES ? 0 : (int)(threadIdx.x >> 6), (TB_) == LANES ? 1 : (int)(blockDim.x >> 6))

// f_relax / c_relax / forward_solve (mgrit.py:292-370,459-486). ROLE only separates the launches by purpose (distinct
// kernel symbols in rocprof traces; the weighted C-relaxation is the only one that re-reads the old u_i).
// ROLE_FC: f_relax followed by c_relax (mgrit.py:270-275 on a level the FAS sweep of the finer level has just filled) in one
// pass: a run = the F-points of an interval AND the C-point that closes it; the starting C-point is read from v (right after
// the restriction v == u bit for bit, and v is not written by this pass, so no run sees a C-point its neighbour has already
// relaxed); only the closing C-point is stored -- the F-points of the way down are read by nobody (the F-relaxation that
// follows is folded into the FAS pass, fas_fused1_kernel PROP, and the way up rewrites them).
enum { ROLE_F = 0, ROLE_C = 1, ROLE_C_WEIGHTED = 2, ROLE_FC = 3 };

// TB (the sweeps of a Heat1D cycle: relax_kernel ROLE_FC, ecf_kernel, cfas_kernel, ecfr_kernel, fas_fused1_kernel): the workgroup
// size the instance is compiled for. States of one group (n <= 1024) run one WAVE per state: nothing hides behind other waves, the
// kernel's time is that wave's chain of dependent instructions. Compiled for 1024 threads the kernels keep to 128 VGPRs and spill
// 100-350 SGPRs and up to 90 VGPRs, whose reloads sit in that chain in every Phi; the TB = 64 instances (512 VGPRs, no VGPR spill)
// take the launches of such levels and know at compile time that the state is ONE group (heat_solve<.., ONE>): config 2's five
// sweeps 17-32 us -> 11-20 us each. Same source; the values of the 1024-thread instances (MGRIT_HIP_SMALL_WG=0 launches those
// everywhere), up to the sign of a zero.
bool small_wg_instances() {
    const char *s = std::getenv("MGRIT_HIP_SMALL_WG");      // (read per launch: a test compares the two families in one process)
    return !(s && s[0] == '0' && s[1] == 0);
}
// ... and TB = 512 for states of up to 8192 values (workgroups of 128 .. 512 threads): 256 VGPRs, no VGPR spills -- these sweeps are
// bound by memory bandwidth and by Phi's issue slots, and spill traffic costs both (heat_1d 8192 x 16385: 1.14 -> 1.01 ms per cycle).
// MGRIT_HIP_MID_WG=0: the 1024-thread instances for them.
bool mid_wg_instances() {
    const char *s = std::getenv("MGRIT_HIP_MID_WG");
    return !(s && s[0] == '0' && s[1] == 0);
}
// the instance (its TB) that takes a launch of T threads
int sweep_tb(int T) { return T == LANES ? (small_wg_instances() ? LANES : 1024) : T <= 512 ? (mid_wg_instances() ? 512 : 1024) : 1024; }

template <int KIND, int FORCE, bool USE_G, int ROLE, int TB = 1024>
__global__ void __launch_bounds__(TB) relax_kernel(LevelDev L, const int32_t *__restrict__ run_start,
                                                     const int32_t *__restrict__ run_len, int n_runs, double w, double w1) {
    WG_PROLOGUE_TB(TB);
    constexpr bool ONE = TB == LANES;   // one wave per state: Phi without the cross-group exchange (heat_solve)
    // persistent workgroups: the grid is sized to the chip (not to the run list), each workgroup walks the runs with
    // stride gridDim.x and keeps the coefficient tables of its current time-step size in LDS / SGPRs across runs
    for (wq.begin(L.sched, L.xcc0_limit, wgq_slot, t); wq.cur < n_runs; wq.advance(t)) {
        const int r = wq.cur;
        wq.prefetch(t);
        const int start = run_start[r], len = run_len[r];
        double x[E], gi[E];
        load_row_nt((ROLE == ROLE_FC ? L.v : L.u) + (size_t)(start - 1) * L.ld, sl, x, ROLE == ROLE_FC ? L.stream_rows : 0);
        for (int i = start; i < start + len; ++i) {
            if (USE_G) load_row_nt(L.g + (size_t)i * L.ld, sl, gi, ROLE == ROLE_FC ? L.stream_rows : 0);  // in flight while Phi runs
            phi_apply<KIND, FORCE, false, 0, false, ONE>(x, ctx, L, i, sm, t, lane, wave, G);
            if (USE_G) {
#pragma unroll
                for (int k = 0; k < E; ++k) x[k] = gi[k] + x[k];
            }
            if (ROLE == ROLE_C_WEIGHTED) {
                double uo[E];
                load_row(L.u + (size_t)i * L.ld, sl, uo);
#pragma unroll
                for (int k = 0; k < E; ++k) x[k] = x[k] * w + uo[k] * w1;
            }
            if (ROLE != ROLE_FC || i == start + len - 1) store_row(L.u + (size_t)i * L.ld, sl, x);
        }
    }
    wq.end(t);
}

// error_correction + f_relax in one pass for the identity transfer (mgrit.py:715-726 followed by 292-333): a run whose
// predecessor is a corrected C-point applies the correction itself -- u_c = u_c + (u^{l+1}_j - v^{l+1}_j), same operation
// order as interp_rows_kernel, with v^{l+1}_j taken from u_c itself (see below) -- writes the C-point back and goes on with the F-points, so the C-point travels through HBM
// once instead of twice. ec_coarse[r] = coarse slot j of run r's predecessor, or -1 (ghost / uncorrected predecessor).
template <int KIND, int FORCE, bool USE_G, int TB = 1024>
__global__ void __launch_bounds__(TB) ecf_kernel(LevelDev L, LevelDev Lc, const int32_t *__restrict__ run_start,
                                                   const int32_t *__restrict__ run_len, const int32_t *__restrict__ ec_coarse,
                                                   int n_runs) {
    WG_PROLOGUE_TB(TB);
    constexpr bool ONE = TB == LANES;   // one wave per state: Phi without the cross-group exchange (heat_solve)
    for (wq.begin(L.sched, L.xcc0_limit, wgq_slot, t); wq.cur < n_runs; wq.advance(t)) {
        const int r = wq.cur;
        wq.prefetch(t);
        const int start = run_start[r], len = run_len[r], j = ec_coarse[r];
        double x[E], gi[E];
        load_row_nt(L.u + (size_t)(start - 1) * L.ld, sl, x, L.stream_rows);
        if (j >= 0) {
            // v^{l+1}_j is not read: fas_residual made it the clone of u^l at this very C-point (identity transfer) and nothing
            // has touched level l since (iteration(): fas_residual(l), iteration(l+1), then this), so v_j == x bit for bit
            double uc[E];
            load_row_nt(Lc.u + (size_t)j * Lc.ld, sl, uc, Lc.stream_rows);
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = x[k] + (uc[k] - x[k]);
            store_row_nt(L.u + (size_t)(start - 1) * L.ld, sl, x, L.stream_rows);
        }
        for (int i = start; i < start + len; ++i) {
            if (USE_G) load_row_nt(L.g + (size_t)i * L.ld, sl, gi, L.stream_rows);
            phi_apply<KIND, FORCE, false, 0, false, ONE>(x, ctx, L, i, sm, t, lane, wave, G);
            if (USE_G) {
#pragma unroll
                for (int k = 0; k < E; ++k) x[k] = gi[k] + x[k];
            }
            store_row_nt(L.u + (size_t)i * L.ld, sl, x, L.stream_rows);
        }
    }
    wq.end(t);
}

// AtMgrit.forward_solve (core/at_mgrit.py:79-87): point p of the coarsest level is recomputed from the OLD value k-1
// points back by k-1 steps with the FAS right-hand side; all points are independent (that is the point of AT-MGRIT: no
// sequential coarsest-level solve). u_old = a copy of the u slab taken before the launch.
template <int KIND, int FORCE>
__global__ void __launch_bounds__(1024) at_kernel(LevelDev L, const double *__restrict__ u_old, int k) {
    WG_PROLOGUE;
    for (int p = 1 + blockIdx.x; p < L.n_pts; p += gridDim.x) {
        const int s = p - k + 1 > 0 ? p - k + 1 : 0;
        double x[E], gi[E];
        load_row(u_old + (size_t)s * L.ld, sl, x);
        for (int i = s + 1; i <= p; ++i) {
            load_row(L.g + (size_t)i * L.ld, sl, gi);
            phi_apply<KIND, FORCE>(x, ctx, L, i, sm, t, lane, wave, G);
#pragma unroll
            for (int q = 0; q < E; ++q) x[q] = gi[q] + x[q];
        }
        store_row(L.u + (size_t)p * L.ld, sl, x);
    }
}

#include "mgrit_hip_chain.inc"

// compute_residual (mgrit.py:387-413): out[run] = || Phi(u_{i-1}) - u_i ||^2
template <int KIND, int FORCE>
__global__ void __launch_bounds__(1024) residual_kernel(LevelDev L, const int32_t *__restrict__ run_start, int n_runs,
                                                        double *__restrict__ out) {
    WG_PROLOGUE;
    for (wq.begin(L.sched, L.xcc0_limit, wgq_slot, t); wq.cur < n_runs; wq.advance(t)) {  // persistent workgroups, tables stay resident
        const int r = wq.cur;
        wq.prefetch(t);
        const int i = run_start[r];
        double x[E], ui[E];
        load_row(L.u + (size_t)(i - 1) * L.ld, sl, x);
        phi_apply<KIND, FORCE>(x, ctx, L, i, sm, t, lane, wave, G);
        // u_i is loaded AFTER Phi: fetched ahead it is a second live vector in a 128-VGPR workgroup, the kernel spills
        // (84 B per lane of scratch traffic) and runs 25 % longer than with the load latency exposed
        load_row(L.u + (size_t)i * L.ld, sl, ui);
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = x[k] - ui[k];
        const double tot = block_sumsq(x, sm, L.n, t, lane, wave, G);
        if (t == 0) out[r] = tot;
    }
    wq.end(t);
}

// compute_jump (mgrit.py:372-385): out[run] = || u_i - prev_i ||^2
__global__ void __launch_bounds__(1024) jump_kernel(LevelDev L, const int32_t *__restrict__ run_start,
                                                    const double *__restrict__ prev, double *__restrict__ out) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, G = blockDim.x >> 6;
    const Smem sm = carve_smem(smem_raw, L.T, L.kind != MGRIT_HIP_STEPPER_ADVECTION1D);
    const unsigned sl = slot0(t);
    const int i = run_start[blockIdx.x];
    double x[E], p[E];
    load_row(L.u + (size_t)i * L.ld, sl, x);
    load_row(prev + (size_t)i * L.ld, sl, p);
#pragma unroll
    for (int k = 0; k < E; ++k) x[k] = x[k] - p[k];
    const double tot = block_sumsq(x, sm, L.n, t, lane, wave, G);
    if (t == 0) out[blockIdx.x] = tot;
}

// fas_residual, fine half (mgrit.py:528-532 / 538-543): out_p = Phi_l(u_{i-1}) - u_i   or   (g_i - u_i) + Phi_l(u_{i-1})
template <int KIND, int FORCE>
__global__ void __launch_bounds__(1024) fas_fine_kernel(LevelDev L, const int32_t *__restrict__ fine_idx,
                                                        const int32_t *__restrict__ out_idx, double *__restrict__ out,
                                                        int out_ld, int use_g) {
    WG_PROLOGUE;
    const int i = fine_idx[blockIdx.x];
    double x[E], ui[E];
    load_row(L.u + (size_t)(i - 1) * L.ld, sl, x);
    load_row(L.u + (size_t)i * L.ld, sl, ui);
    if (use_g) {  // fold (g_i - u_i) first: one live vector less while Phi runs
        double gi[E];
        load_row(L.g + (size_t)i * L.ld, sl, gi);
#pragma unroll
        for (int k = 0; k < E; ++k) ui[k] = gi[k] - ui[k];
    }
    phi_apply<KIND, FORCE>(x, ctx, L, i, sm, t, lane, wave, G);
    if (use_g) {
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = ui[k] + x[k];
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = x[k] - ui[k];
    }
    store_row(out + (size_t)out_idx[blockIdx.x] * out_ld, sl, x);
}

// fas_residual, coarse half (mgrit.py:533-536 / 544-547): g_j = (g_j + v_j) - Phi_{l+1}(v_{j-1})
template <int KIND, int FORCE>
__global__ void __launch_bounds__(1024) fas_coarse_kernel(LevelDev L, const int32_t *__restrict__ coarse_idx, int have_sum) {
    // have_sum: the row of g already holds g_j + v_j (gen_down_kernel adds the two in registers)
    WG_PROLOGUE;
    const int j = coarse_idx[blockIdx.x];
    double x[E], a[E];
    load_row(L.v + (size_t)(j - 1) * L.ld, sl, x);
    load_row(L.g + (size_t)j * L.ld, sl, a);
    if (!have_sum) {
        double b[E];
        load_row(L.v + (size_t)j * L.ld, sl, b);
#pragma unroll
        for (int k = 0; k < E; ++k) a[k] = a[k] + b[k];
    }
    phi_apply<KIND, FORCE>(x, ctx, L, j, sm, t, lane, wave, G);
#pragma unroll
    for (int k = 0; k < E; ++k) x[k] = a[k] - x[k];
    store_row(L.g + (size_t)j * L.ld, sl, x);
}

// fas_residual fused for the identity spatial transfer (GridTransferCopy): per C-point j >= 1 with fine slot i, previous
// fine C slot ip and coarse slot j (mgrit.py:498-500,520,524-547):
//   u^{l+1}_j = v^{l+1}_j = u^l_i ;  g^{l+1}_j = ((Phi_l(u^l_{i-1}) - u^l_i) + v_j) - Phi_{l+1}(v_{j-1}),  v_{j-1} = u^l_{ip}
// (lvl > 0: (g^l_i - u^l_i) + Phi_l(u^l_{i-1})). Persistent workgroups in two phases, so that each level's coefficient tables
// are loaded into LDS once per workgroup instead of twice per C-point: phase 1 applies the fine Phi to all of the
// workgroup's points (writes u, v and the partial g of the coarse level), phase 2 the coarse Phi (reads u^l_{ip} and the
// partial g it wrote itself). 4-5 vectors read, 4 written per C-point instead of 11-12.
template <int KIND, int FORCE, int TB = 1024>   // (TB = 512: the instance for states of up to 8192 values, sweep_tb; 46-100 spilled VGPRs otherwise)
__global__ void __launch_bounds__(TB) fas_fused_kernel(LevelDev L, LevelDev Lc, const int32_t *__restrict__ fine_idx,
                                                         const int32_t *__restrict__ prev_idx,
                                                         const int32_t *__restrict__ coarse_idx, int n_items, int use_g) {
    WG_PROLOGUE;
    for (int p = blockIdx.x; p < n_items; p += gridDim.x) {
        const int i = fine_idx[p], j = coarse_idx[p];
        double x[E], w[E];
        load_row(L.u + (size_t)(i - 1) * L.ld, sl, x);
        load_row(L.u + (size_t)i * L.ld, sl, w);
        store_row(Lc.u + (size_t)j * Lc.ld, sl, w);
        store_row(Lc.v + (size_t)j * Lc.ld, sl, w);
        if (use_g) {
            double gi[E];
            load_row(L.g + (size_t)i * L.ld, sl, gi);
#pragma unroll
            for (int k = 0; k < E; ++k) w[k] = gi[k] - w[k];
        }
        phi_apply<KIND, FORCE>(x, ctx, L, i, sm, t, lane, wave, G);
        if (use_g) {
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = w[k] + x[k];
            load_row(L.u + (size_t)i * L.ld, sl, w);   // u^l_i once more (one live vector less while Phi runs)
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k) x[k] = x[k] - w[k];
        }
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = x[k] + w[k];   // + v_j
        store_row(Lc.g + (size_t)j * Lc.ld, sl, x);
    }
    {
        const int par = ctx.parity;
        ctx_init<KIND, FORCE>(ctx, Lc, t);       // coarse level: its own forcing factor and coefficient sets
        ctx.parity = par;
    }
    for (int p = blockIdx.x; p < n_items; p += gridDim.x) {
        const int ip = prev_idx[p], j = coarse_idx[p];
        double x[E], w[E];
        load_row(L.u + (size_t)ip * L.ld, sl, w);   // v_{j-1}
        load_row(Lc.g + (size_t)j * Lc.ld, sl, x);  // partial g, written above by this very lane
        phi_apply<KIND, FORCE>(w, ctx, Lc, j, sm, t, lane, wave, G);
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = x[k] - w[k];
        store_row(Lc.g + (size_t)j * Lc.ld, sl, x);
    }
}

// Phi of ANOTHER Heat1D level (Lc, step j) applied to w while the calling kernel's own level keeps its tables in LDS: the
// correction table, Pt and the forcing factors of Lc are read straight from global memory -- 144 KB per level that every
// workgroup reads, so they stay in L2 and cost no HBM traffic -- and its scalar coefficients live only for this call (the
// caller reloads its own set afterwards: both at once do not fit the SGPR file). Arithmetic identical to phi_apply on Lc.
template <int FORCE, bool LDSBAR = false, bool ONE = false>
__device__ __forceinline__ void phi_other_level(double (&w)[E], StepCtx &ctx, const LevelDev &Lc, int j, const Smem &sm, unsigned sl,
                                                int t, int lane, int wave, int G) {
    const int cj = Lc.one_cset ? 0 : ld_uniform(Lc.cidx + j);
    const CSet *gc = Lc.cs + cj;
    Smem smc = sm;
    // (a level with ONE coefficient set: the caller has staged its lane powers and group factors in LDS, stage_other_level --
    // read from global memory they are vector loads, which retire behind every row store the wave still has in flight)
    smc.wf = Lc.one_cset ? sm.wf2 : const_cast<double *>(gc->pg);
    smc.lp = Lc.one_cset ? sm.lp2 : const_cast<double *>(gc->lp);
    smc.pt = const_cast<double2 *>(Lc.ptP) + (size_t)cj * 1024;
    if (FORCE != 0) {
        for (int kk = 0; kk < Lc.K; ++kk) {
            const double ck = ld_uniform(Lc.tc + (size_t)kk * Lc.n_pts + j);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double2 sv = Lc.sP[(size_t)kk * 8 * Lc.T + sl + q * 64];
                w[2 * q] = fma(sv.x, ck, w[2 * q]);
                w[2 * q + 1] = fma(sv.y, ck, w[2 * q + 1]);
            }
        }
    }
    // the other level's scalars take the PLACE of the caller's (ctx.c): with a second set beside it ~120 wave-uniform doubles
    // would be live at once, more than the SGPR file holds. The caller reloads its own set afterwards, unconditionally, so
    // that the compiler sees ctx.c dead across this call.
    load_coef(ctx.c, gc);
    const LaneCoef lcc = lane_coef(smc.lp, lane);
    const int par = ctx.parity;
    ctx.parity ^= 1;
    heat_solve<true, LDSBAR, ONE>(w, ctx.c, lcc, smc, sm.ga + par * MAX_G, sm.gb + par * MAX_G, Lc.n, t, lane, wave, G);
}

__device__ __forceinline__ void stage_other_level(const Smem &sm, const LevelDev &Lc, int t) {
    if (Lc.one_cset) {
        if (t < LANES) sm.lp2[t] = Lc.cs->lp[t];
        if (t < 3 * MAX_G) sm.wf2[t] = (&Lc.cs->pg[0])[t];   // pg | qg | qg2 are consecutive members
        __syncthreads();
    }
}

// Interval list of the fused level sweeps below: item = the interval from one C-point to the next,
//   cstart / cend = fine slots of the two C-points (at least one F-point between them), cend_coarse = coarse slot of the
//   closing C-point, res_pos = position of the closing C-point in the residual output.
// Workgroups take CHUNKS of consecutive intervals (chunk_first, chunk_len) and walk them in time order, so the C-point an
// interval ends on is the one the next interval starts from and stays in registers; chunk_start_coarse = coarse slot of the
// chunk's first C-point when that point takes part in the sweep (it is C-relaxed / corrected), -1 when it is not (the first
// point of the time grid).
// keep[i] says which rows of the coarse level the closing C-point of interval i really needs (mgrit_hip_intervals_create):
//   bit 0: u^{l+1} -- not needed at a coarse F-point (the coarse level's first sweep, an F-relaxation, writes it before anything
//          reads it) nor anywhere on a coarsest level that is solved by forward_solve;
//   bit 1: v^{l+1} -- its only reader is the error correction, which may take the same bits from the fine C-point (identity
//          transfer); needed where a chunk of the way up starts (that row is being overwritten by the chunk in front of it).
struct IntervalsDev {
    const int32_t *cstart, *cend, *cend_coarse, *res_pos, *chunk_first, *chunk_len, *chunk_start_coarse, *keep;
    int n_chunks;
};

// c_relax + f_relax + fas_residual of one level in ONE pass (mgrit.py:335-370, 292-333, 488-549 as Mgrit.iteration calls them
// one after the other, mgrit.py:277-281), Heat1D on both levels, identity transfer, weight 1. Per interval (C_j, C_{j+1}]:
//   q        = Phi_{l+1}(C'_j)                      parked in LDS: every Phi of this kernel evaluates its rank-one correction in
//                                                   closed form (heat_solve<CLOSED>), so the table's 8 B per DOF of LDS are free
//   F'_last  = Phi_l^{m-1}(C'_j)                    the F-relaxation; its points are NOT stored: nothing reads a level's F-points
//                                                   before the error correction + F-relaxation on the way up rewrites them
//   C'_{j+1} = Phi_l(u_old[c_{j+1} - 1])            the C-relaxation, from the OLD last F-point of the interval (F rows are
//                                                   never written here, so no other workgroup can have touched it). With
//                                                   pre = 1 that row already HOLDS Phi_l(last F-point): the residual check of
//                                                   the cycle before computed exactly this value (r = Phi(u_{i-1}) - u_i,
//                                                   mgrit.py:401-405) and ecfr_kernel (store_f = 2) left it there
//   u^{l+1}_{j+1} = v^{l+1}_{j+1} = C'_{j+1},   g^{l+1}_{j+1} = ((Phi_l(F'_last) - C'_{j+1}) + C'_{j+1}) - q
// Rows through HBM per interval: 1 read + 3..5 written (IntervalsDev::keep) instead of 12 (C-relax 2, F-relax m, fused FAS 6). A chunk's first C-point is recomputed from its old F-point (one more Phi per chunk), not waited for.
// (The /*STAMP n*/ and /*DRAIN*/ comments mark the phases for tools/cfas_timeline.py, which turns them into wall-clock stamps in an
// experiment build of its own: profiles/r04_cfas_timeline.txt. Per interval of config 3: five Phi at ~2.1 us of solve each plus
// ~1.7 us each for streaming the forcing factor from L2 -- LDS, where ecfr_kernel keeps it, holds q here --, 4.4 us of row
// traffic that the wave waits for, 2 us until the stores of the interval before have drained.)
template <int FORCE, int TB = 1024>
__global__ void __launch_bounds__(TB) cfas_kernel(LevelDev L, LevelDev Lc, IntervalsDev I, int pre) {
    constexpr int KIND = MGRIT_HIP_STEPPER_HEAT1D;
    WG_PROLOGUE_TB(TB);
    constexpr bool ONE = TB == LANES;   // one wave per state: Phi without the cross-group exchange (heat_solve)
    stage_other_level(sm, Lc, t);
    for (wq.begin(L.sched, L.xcc0_limit, wgq_slot, t); wq.cur < I.n_chunks; wq.advance(t)) {
        const int k = wq.cur;
        wq.prefetch(t);
        const int i0 = I.chunk_first[k], cnt = I.chunk_len[k];
        double x[E];
        {
            const int cs = I.cstart[i0];
            if (I.chunk_start_coarse[k] >= 0) {   // C'_j of the chunk's first C-point, recomputed (pre: it is what the row holds)
                load_row_nt(L.u + (size_t)(cs - 1) * L.ld, sl, x, L.stream_rows);
                if (!pre) {
                    if (ctx.cur >= 0) load_coef(ctx.c, L.cs + ctx.cur);
                    phi_apply<KIND, FORCE, true, 0, false, ONE>(x, ctx, L, cs, sm, t, lane, wave, G);
                }
            } else {
                load_row_nt(L.u + (size_t)cs * L.ld, sl, x, L.stream_rows);
            }
        }
        /*STAMP 6*/
        for (int it = i0; it < i0 + cnt; ++it) {
            const int cs = I.cstart[it], ce = I.cend[it], jc = I.cend_coarse[it];
            /*DRAIN*/ /*STAMP 8*/
            {   // q = Phi_{l+1}(v_j), v_j = C'_j
                double w[E];
#pragma unroll
                for (int e = 0; e < E; ++e) w[e] = x[e];
                phi_other_level<FORCE, false, ONE>(w, ctx, Lc, jc, sm, sl, t, lane, wave, G);
#pragma unroll
                for (int q = 0; q < 8; ++q) sm.tab[sl + q * 64] = make_double2(w[2 * q], w[2 * q + 1]);   // parked in LDS
            }
            /*STAMP 0*/
            load_coef(ctx.c, L.cs + (ctx.cur >= 0 ? ctx.cur : 0));   // back to this level's set (ctx.cur < 0: any set, phi_apply loads the right one)
            /*STAMP 9*/
            for (int i = cs + 1; i < ce; ++i) phi_apply<KIND, FORCE, true, 0, false, ONE>(x, ctx, L, i, sm, t, lane, wave, G);
            /*STAMP 1*/
            double b[E];
            load_row_nt(L.u + (size_t)(ce - 1) * L.ld, sl, b, L.stream_rows);   // (requested one Phi earlier it costs more in spills than it hides)
            /*STAMP 2*/
            if (!pre) phi_apply<KIND, FORCE, true, 0, false, ONE>(b, ctx, L, ce, sm, t, lane, wave, G);
            // the residual Phi in two parts around the row stores: its forcing term (vector loads of the space factor) AHEAD of
            // them -- a load issued behind stores retires behind them (vmcnt counts in order), and the Phi would wait for their
            // round trip to HBM --, its solve (LDS and registers only) behind them
            phi_apply<KIND, FORCE, true, 1, false, ONE>(x, ctx, L, ce, sm, t, lane, wave, G);
            store_row_nt(L.u + (size_t)ce * L.ld, sl, b, L.stream_rows);
            const int keep = __builtin_amdgcn_readfirstlane(I.keep[it]);
            if (keep & 1) store_row_nt(Lc.u + (size_t)jc * Lc.ld, sl, b, Lc.stream_rows);
            if (keep & 2) store_row_nt(Lc.v + (size_t)jc * Lc.ld, sl, b, Lc.stream_rows);
            /*STAMP 3*/
            phi_apply<KIND, FORCE, true, 2, false, ONE>(x, ctx, L, ce, sm, t, lane, wave, G);
            /*STAMP 4*/
#pragma unroll
            for (int e = 0; e < E; ++e) x[e] = x[e] - b[e];
#pragma unroll
            for (int e = 0; e < E; ++e) x[e] = x[e] + b[e];
#pragma unroll
            for (int q = 0; q < 8; ++q) {   // q, parked above by this very lane
                const double2 w = sm.tab[sl + q * 64];
                x[2 * q] = x[2 * q] - w.x;
                x[2 * q + 1] = x[2 * q + 1] - w.y;
            }
            store_row_nt(Lc.g + (size_t)jc * Lc.ld, sl, x, Lc.stream_rows);
#pragma unroll
            for (int e = 0; e < E; ++e) x[e] = b[e];
            /*STAMP 5*/
        }
    }
    wq.end(t);
    /*STAMP 7*/
}

// (USE_G, !RES: the same pass on a coarser level -- error_correction + f_relax with the rows of g, every F-point stored, no
// residual. Unlike ecf_kernel, whose interval corrects the C-point it STARTS from, an interval here corrects the C-point it
// ends on: in a planned cycle a block of time points then needs the coarsest-level chain of ITS block only.)
// error_correction + f_relax + compute_residual of level 0 in ONE pass (mgrit.py:715-726, 292-333 as Mgrit.iteration calls
// them, mgrit.py:283-284, then 387-413 from convergence_criterion), identity transfer. Per interval (C_j, C_{j+1}]:
//   F''      = Phi-propagation from the corrected C''_j, stored -- all of them (store_f = 1), or with store_f = 0 only the last
//              one (the point the next C-relaxation starts from): F-points are a function of the C-points, and an F-relaxation
//              (mgrit_hip_relax mode F) rebuilds them bit for bit whenever somebody wants to see them; store_f = 2 goes one
//              step further and leaves Phi_l(F''_last) -- computed here for the residual -- in the last F-point's row: that IS
//              the value the next cycle's C-relaxation assigns, so mgrit_hip_cf_fas(pre = 1) takes it without its own Phi
//   C''_{j+1} = v^{l+1}_{j+1} + (u^{l+1}_{j+1} - v^{l+1}_{j+1})   (v^{l+1}_{j+1} IS u^l at that C-point, bit for bit: read from the
//                                                                 fine row inside a chunk, from v -- which nobody writes on the
//                                                                 way up -- for the C-point a chunk starts from), stored
//   out[res_pos] = || Phi_l(F''_last) - C''_{j+1} ||^2
// 2 rows read + m written per interval instead of 2 + m (correction + F-relaxation) and 2 more for the residual.
template <int FORCE, bool USE_G, bool RES, int TB = 1024>
__global__ void __launch_bounds__(TB) ecfr_kernel(LevelDev L, LevelDev Lc, IntervalsDev I, double *__restrict__ out, int store_f,
                                                    double *const *__restrict__ mirror, int mirror_row0) {
    constexpr int KIND = MGRIT_HIP_STEPPER_HEAT1D;
    constexpr bool CF = FORCE == 4;   // forcing factor in LDS, so every Phi in closed form
    WG_PROLOGUE_TB(TB);
    constexpr bool ONE = TB == LANES;   // one wave per state: Phi without the cross-group exchange (heat_solve)
    stage_forcing<FORCE>(sm, L, sl);
    // C-point mirror (mgrit_hip_cpoint_mirror): every corrected C-point also goes to row mirror_row0 + res_pos of the slab the
    // caller has named for THIS cycle (read once per workgroup: the caller changes it between cycles, on the stream)
    double *const mir = (RES && mirror) ? ld_uniform_ptr(mirror) : nullptr;
    for (wq.begin(L.sched, L.xcc0_limit, wgq_slot, t); wq.cur < I.n_chunks; wq.advance(t)) {
        const int k = wq.cur;
        wq.prefetch(t);
        const int i0 = I.chunk_first[k], cnt = I.chunk_len[k];
        double x[E];
        {
            const int cs = I.cstart[i0], js = I.chunk_start_coarse[k];
            if (js >= 0) {   // the corrected value of the chunk's first C-point (stored by the chunk that ends on it)
                double w[E];
                load_row_nt(Lc.v + (size_t)js * Lc.ld, sl, x, Lc.stream_rows);
                load_row_nt(Lc.u + (size_t)js * Lc.ld, sl, w, Lc.stream_rows);
#pragma unroll
                for (int e = 0; e < E; ++e) x[e] = x[e] + (w[e] - x[e]);
            } else {
                load_row_nt(L.u + (size_t)cs * L.ld, sl, x, L.stream_rows);
            }
        }
        for (int it = i0; it < i0 + cnt; ++it) {
            const int cs = I.cstart[it], ce = I.cend[it], jc = I.cend_coarse[it];
            for (int i = cs + 1; i < ce; ++i) {
                double gi[E];
                if (USE_G) load_row_nt(L.g + (size_t)i * L.ld, sl, gi, L.stream_rows);   // in flight while Phi runs
                phi_apply<KIND, FORCE, CF, 0, false, ONE>(x, ctx, L, i, sm, t, lane, wave, G);
                if (USE_G) {
#pragma unroll
                    for (int e = 0; e < E; ++e) x[e] = gi[e] + x[e];
                }
                if (store_f == 1 || (store_f == 0 && i == ce - 1)) store_row_nt(L.u + (size_t)i * L.ld, sl, x, L.stream_rows);
            }
            double b[E];
            {   // v^{l+1}_{j+1} from the fine row itself (the same bits; only this chunk writes that row), see IntervalsDev::keep
                double w[E];
                load_row_nt(L.u + (size_t)ce * L.ld, sl, b, L.stream_rows);
                load_row_nt(Lc.u + (size_t)jc * Lc.ld, sl, w, Lc.stream_rows);
#pragma unroll
                for (int e = 0; e < E; ++e) b[e] = b[e] + (w[e] - b[e]);
            }
            store_row_nt(L.u + (size_t)ce * L.ld, sl, b, L.stream_rows);
            if (RES && mir) store_row_nt(mir + (size_t)(mirror_row0 + I.res_pos[it]) * L.ld, sl, b, 1);
            if (RES) {
                phi_apply<KIND, FORCE, CF, 0, false, ONE>(x, ctx, L, ce, sm, t, lane, wave, G);
                if (store_f == 2) store_row_nt(L.u + (size_t)(ce - 1) * L.ld, sl, x, L.stream_rows);   // Phi(last F-point): the next C-relaxation's value
#pragma unroll
                for (int e = 0; e < E; ++e) x[e] = x[e] - b[e];
                const double tot = block_sumsq(x, sm, L.n, t, lane, wave, G);
                if (t == 0) out[I.res_pos[it]] = tot;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) x[e] = b[e];
        }
    }
    wq.end(t);
}

// The same sweep in ONE pass per C-point for Heat1D (both levels of the pair): fine Phi with the fine level's tables in LDS
// as everywhere else, then the coarse Phi with the coarse level's correction table, Pt and forcing factors read straight
// from global memory -- 144 KB per level that every workgroup reads, so they stay in L2 and cost no HBM traffic -- and the
// scalar coefficients of the level in use reloaded in front of each Phi (scalar loads; both sets at once would not fit the
// SGPR file). The partial g of the two-phase form never leaves the registers: 3 vectors read (+g_i), 3 written per C-point
// instead of 4-5 and 4. Arithmetic identical to fas_fused_kernel.
template <int FORCE, bool PROP, int TB = 1024>
__global__ void __launch_bounds__(TB) fas_fused1_kernel(LevelDev L, LevelDev Lc, const int32_t *__restrict__ fine_idx,
                                                          const int32_t *__restrict__ prev_idx,
                                                          const int32_t *__restrict__ coarse_idx, int n_items, int use_g,
                                                          int opts) {
    // PROP (opts bit 0): the F-relaxation in front of the sweep (mgrit.py:275 / 271) is part of it -- the F-points between the
    //   previous C-point ip and i are stepped through here, u_k = g_k + Phi(u_{k-1}), and not stored (nobody reads them before
    //   the way up rewrites them); bit 1: u^{l+1}_j is not stored (a coarsest level that forward_solve overwrites unread)
    constexpr int KIND = MGRIT_HIP_STEPPER_HEAT1D;
    constexpr bool CF = FORCE == 4;   // forcing factor in LDS, so every Phi in closed form
    WG_PROLOGUE_TB(TB);
    constexpr bool ONE = TB == LANES;   // one wave per state: Phi without the cross-group exchange (heat_solve)
    stage_forcing<FORCE>(sm, L, sl);
    stage_other_level(sm, Lc, t);
    Smem smc = sm;
    for (wq.begin(L.sched, L.xcc0_limit, wgq_slot, t); wq.cur < n_items; wq.advance(t)) {
        const int p = wq.cur;
        wq.prefetch(t);
        const int i = fine_idx[p], j = coarse_idx[p], ip = prev_idx[p];
        double x[E], w[E];
        if (ctx.cur >= 0) load_coef(ctx.c, L.cs + ctx.cur);   // not kept alive across the coarse Phi below
        if (PROP) {
            // ONE call site of the fine Phi for the F-steps and for the step onto the C-point (two inlined copies made the
            // register allocator spill 250 VGPRs): every step is x = w + Phi(x) with w = g_k, and w = g_i - u_i for the last
            row_load2(L.u + (size_t)ip * L.ld, sl, x, L.stream_rows);
            for (int k = ip + 1; k <= i; ++k) {
                row_load2(L.g + (size_t)k * L.ld, sl, w, L.stream_rows);   // in flight while Phi runs (PROP implies use_g)
                if (k == i) {
                    double ui[E];
                    row_load2(L.u + (size_t)i * L.ld, sl, ui, 0);
                    if (!(opts & 2)) row_store2(Lc.u + (size_t)j * Lc.ld, sl, ui, Lc.stream_rows);
                    row_store2(Lc.v + (size_t)j * Lc.ld, sl, ui, Lc.stream_rows);
#pragma unroll
                    for (int e = 0; e < E; ++e) w[e] = w[e] - ui[e];
                }
                phi_apply<KIND, FORCE, CF, 0, false, ONE>(x, ctx, L, k, sm, t, lane, wave, G);
#pragma unroll
                for (int e = 0; e < E; ++e) x[e] = w[e] + x[e];
            }
            row_load2(L.u + (size_t)i * L.ld, sl, w, 0);   // u^l_i once more (one live vector less while Phi runs)
        } else {
            row_load2(L.u + (size_t)(i - 1) * L.ld, sl, x, 0);
            row_load2(L.u + (size_t)i * L.ld, sl, w, 0);
            if (!(opts & 2)) row_store2(Lc.u + (size_t)j * Lc.ld, sl, w, Lc.stream_rows);
            row_store2(Lc.v + (size_t)j * Lc.ld, sl, w, Lc.stream_rows);
            if (use_g) {
                double gi[E];
                row_load2(L.g + (size_t)i * L.ld, sl, gi, 0);
#pragma unroll
                for (int k = 0; k < E; ++k) w[k] = gi[k] - w[k];
            }
            phi_apply<KIND, FORCE, CF, 0, false, ONE>(x, ctx, L, i, sm, t, lane, wave, G);
            if (use_g) {
#pragma unroll
                for (int k = 0; k < E; ++k) x[k] = w[k] + x[k];
                row_load2(L.u + (size_t)i * L.ld, sl, w, 0);   // u^l_i once more (one live vector less while Phi runs)
            } else {
                row_load2(L.u + (size_t)i * L.ld, sl, w, 0);   // likewise: re-read (an L2 hit) instead of held across Phi
#pragma unroll
                for (int k = 0; k < E; ++k) x[k] = x[k] - w[k];
            }
        }
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = x[k] + w[k];   // + v_j : the partial g of the coarse level
        // ---- coarse Phi on v_{j-1} = u^l_{ip}
        row_load2(L.u + (size_t)ip * L.ld, sl, w, L.stream_rows);
        const int cj = Lc.one_cset ? 0 : ld_uniform(Lc.cidx + j);
        const CSet *gc = Lc.cs + cj;
        smc.wf = Lc.one_cset ? sm.wf2 : const_cast<double *>(gc->pg);   // (staged in LDS: stage_other_level)
        smc.lp = Lc.one_cset ? sm.lp2 : const_cast<double *>(gc->lp);
        smc.pt = const_cast<double2 *>(Lc.ptP) + (size_t)cj * 1024;
        if (FORCE == 4 && (opts & 4)) {   // the coarse level's space factor is the fine level's, bit for bit: the LDS copy
            const double ck = ld_uniform(Lc.tc + j);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double2 sv = sm.tab[sl + q * 64];
                w[2 * q] = fma(sv.x, ck, w[2 * q]);
                w[2 * q + 1] = fma(sv.y, ck, w[2 * q + 1]);
            }
        } else if (FORCE != 0) {
            for (int kk = 0; kk < Lc.K; ++kk) {
                const double ck = ld_uniform(Lc.tc + (size_t)kk * Lc.n_pts + j);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const double2 sv = Lc.sP[(size_t)kk * 8 * Lc.T + sl + q * 64];
                    w[2 * q] = fma(sv.x, ck, w[2 * q]);
                    w[2 * q + 1] = fma(sv.y, ck, w[2 * q + 1]);
                }
            }
        }
        {   // (a set of its own here: in the place of ctx.c, as in phi_other_level, this kernel measured 3 % slower)
            Coef cc;
            load_coef(cc, gc);
            const LaneCoef lcc = lane_coef(smc.lp, lane);
            const int par = ctx.parity;
            ctx.parity ^= 1;
            heat_solve<true, false, ONE>(w, cc, lcc, smc, sm.ga + par * MAX_G, sm.gb + par * MAX_G, Lc.n, t, lane, wave, G);
        }
#pragma unroll
        for (int k = 0; k < E; ++k) x[k] = x[k] - w[k];
        row_store2(Lc.g + (size_t)j * Lc.ld, sl, x, Lc.stream_rows);
    }
    wq.end(t);
}

// --- spatial transfer kernels (bandwidth-bound, elementwise over ROW POSITIONS of the destination) ---------------
// restriction: dst row d_idx[p] <- R(src row s_idx[p]). kind 0 copy (same T: position-wise copy); kind 1 full
// weighting (examples/example_spatial_coarsening.py:33-55: sol[2i]*1/4 + sol[2i+1]*1/2 + sol[2i+2]*1/4).
__global__ void restrict_rows_kernel(const double *__restrict__ src, int src_ld, int T_f, const int32_t *__restrict__ s_idx,
                                     double *__restrict__ dst, int dst_ld, int T_c, const int32_t *__restrict__ d_idx,
                                     int n_c, int kind) {
    const int p = blockIdx.x, pos = blockIdx.y * blockDim.x + threadIdx.x;
    if (pos >= dst_ld) return;
    const double *f = src + (size_t)s_idx[p] * src_ld;
    double *c = dst + (size_t)d_idx[p] * dst_ld;
    if (kind == MGRIT_HIP_TRANSFER_COPY) { c[pos] = f[pos]; return; }
    const int i = row_nat(pos);
    double r = 0.0;
    if (i < n_c) {
        if (kind == MGRIT_HIP_TRANSFER_HEAT1D)
            r = f[row_pos(2 * i)] * 1.0 / 4.0 + f[row_pos(2 * i + 1)] * 1.0 / 2.0 + f[row_pos(2 * i + 2)] * 1.0 / 4.0;
        else {  // periodic full weighting, n_f = 2 n_c
            const int nf = 2 * n_c;
            r = f[row_pos((2 * i - 1 + nf) % nf)] * 1.0 / 4.0 + f[row_pos(2 * i)] * 1.0 / 2.0 + f[row_pos((2 * i + 1) % nf)] * 1.0 / 4.0;
        }
    }
    c[pos] = r;
}

// mode 0: u^l_i = P(u^{l+1}_j)  (mgrit.py:562-563);  mode 1: u^l_i = u^l_i + P(u^{l+1}_j - v^{l+1}_j)  (mgrit.py:724-726)
// linear interpolation of examples/example_spatial_coarsening.py:58-82
__global__ void interp_rows_kernel(double *__restrict__ uf, int f_ld, int T_f, const int32_t *__restrict__ f_idx,
                                   const double *__restrict__ uc, const double *__restrict__ vc, int c_ld, int T_c,
                                   const int32_t *__restrict__ c_idx, int n_f, int n_c, int kind, int mode,
                                   double *__restrict__ rows_out = nullptr, int ld_out = 0) {
    // rows_out (mode 1): the corrected row goes to rows_out[p] instead of back into u^l (mgrit_hip_error_correction_to)
    const int p = blockIdx.x, pos = blockIdx.y * blockDim.x + threadIdx.x;
    if (pos >= f_ld) return;
    double *f = uf + (size_t)f_idx[p] * f_ld;
    const double *e = uc + (size_t)c_idx[p] * c_ld;
    const double *e2 = mode == 1 ? vc + (size_t)c_idx[p] * c_ld : nullptr;
    auto at = [&](int cpos) { return e2 ? e[cpos] - e2[cpos] : e[cpos]; };
    double val;
    if (kind == MGRIT_HIP_TRANSFER_COPY) {
        val = at(pos);
    } else {
        const int j = row_nat(pos);
        if (j >= n_f) return;  // padding stays zero
        if (kind == MGRIT_HIP_TRANSFER_PERIODIC1D) {
            const int i = j >> 1;
            if (j & 1) val = (0.0 + 1.0 / 2.0 * at(row_pos(i))) + 1.0 / 2.0 * at(row_pos((i + 1) % n_c));
            else val = 0.0 + at(row_pos(i));
        } else if (j & 1) val = 0.0 + at(row_pos(j >> 1));
        else {
            const int i = j >> 1;
            val = 0.0;
            if (i - 1 >= 0) val = val + 1.0 / 2.0 * at(row_pos(i - 1));
            if (i < n_c) val = val + 1.0 / 2.0 * at(row_pos(i));
        }
    }
    if (rows_out) rows_out[(size_t)p * ld_out + pos] = f[pos] + val;
    else f[pos] = mode == 0 ? val : f[pos] + val;
}

#include "mgrit_hip_2pts.inc"

#include "mgrit_hip_heat2d.inc"

#include "mgrit_hip_wide.inc"

#include "mgrit_hip_gen.inc"

#include "mgrit_hip_blk.inc"

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct H2DPlan { int count = 0; uint64_t dtbits = 0; int32_t *d_in = nullptr, *d_step = nullptr, *d_dst = nullptr, *d_a = nullptr, *d_b = nullptr; };
struct RunList {
    int n = 0;
    int32_t *d_start = nullptr, *d_len = nullptr, *d_ec = nullptr;  // d_ec: mgrit_hip_ec_runs_create only
    std::vector<int32_t> h_start, h_len;
    std::vector<H2DPlan> h2d_relax, h2d_points;  // Heat2D batch plans (built on first use)
    bool h2d_relax_built = false, h2d_points_built = false;
};
struct PairList {
    int n = 0;
    int32_t *d_fine = nullptr, *d_coarse = nullptr, *d_iota = nullptr, *d_prev = nullptr;
    std::vector<int32_t> h_fine, h_coarse;
    std::vector<H2DPlan> h2d_fine, h2d_coarse, h2d_rows;   // (h2d_rows: the fine half into caller rows, mgrit_hip_fas_fine_rows)
    bool h2d_built = false;
};

struct H2DHost {
    H2DDev dev{};
    std::vector<double> lx, ly, dts;
    std::map<uint64_t, double *> dinv;   // 1/(1 + theta*dt*(lx_k + ly_l)) per distinct dt, [Mi][Mj]
    int HPx = 0, HPy = 0;                // padded half sizes (spectral slots: even modes [0, HP), odd modes [HP, 2 HP))
    double *Fxe = nullptr, *Fxo = nullptr, *FxeT = nullptr, *FxoT = nullptr;   // folded sine tables [HP][HP] and transposes
    double *Fye = nullptr, *Fyo = nullptr, *FyeT = nullptr, *FyoT = nullptr;
    double *W0 = nullptr, *W1 = nullptr, *rowsq = nullptr;
    size_t cap_items = 0;
    // time-parallel forward solve (h2d_block_solve): batch plans of the two passes, the spectra of the block ends, the blocks'
    // propagator tables (one per distinct sequence of step sizes) and a row of zeros (the state every block but the first starts from)
    struct Blk {
        bool built = false;
        int B = 0;
        std::vector<std::vector<H2DPlan>> p1, p3;
        int32_t *d_iota = nullptr, *d_end_all = nullptr, *d_end_tail = nullptr, *d_dsel = nullptr;   // 0 .. B-1; rows e_0 .. e_{B-1}; e_1 .. e_{B-1}; table of block b
        double *spec = nullptr, *Dtab = nullptr, *X = nullptr, *Y = nullptr;   // spectra; propagator tables; the blocks' errors x and inputs u + x
        unsigned *rim_flag = nullptr;   // pinned, device-mapped (theta < 1): set when an error at a block end has a non-zero rim
        hipEvent_t rim_ev = nullptr;
    } blk;
    double *Wc0 = nullptr, *Wc1 = nullptr;   // one item each: the work buffers of the coarsest-level chain, which in a planned
                                             // cycle steps on a second stream BESIDE sweeps that apply this level's Phi too (the
                                             // coarse half of the FAS right-hand side of another block of time points)
};

// Heat1D states of more than MGRIT_HIP_MAX_N values (mgrit_hip_wide.inc): work slabs of the three-launch Phi
struct WideHost {
    double *W = nullptr, *tot = nullptr, *car = nullptr, *z0 = nullptr, *red = nullptr;
    double *T0 = nullptr;   // two-point levels: the new first values of a batch (half rows)
    size_t cap = 0;
};

// Small uploads (index lists, coefficient sets, time factors) without a stream synchronisation: device memory carved from 8 MB
// chunks, the source copied into a pinned mirror of the chunk that lives as long as the engine, the copy enqueued on the engine's
// stream -- ordered in front of every launch that reads it, and the host runs on. (As hipMalloc + hipMemcpyAsync from a
// temporary + hipStreamSynchronize per array, every list created during the first cycle of a solve waited for the sweep that
// was still running on the stream: ~1.3 ms per list, 11 ms per solve of BASELINE config 3, and the constructor's tables waited
// for the zero-fill of the level before.) Pinned chunks are expensive to make (page locking): they go back to a process-wide
// pool when an engine is destroyed -- its stream has been drained by then.
struct UploadArena {
    static constexpr size_t CHUNK = (size_t)8 << 20, SMALL = (size_t)1 << 20;
    struct Chunk { char *dev, *host; };
    std::vector<Chunk> chunks;
    size_t off = CHUNK;
    static std::vector<char *> &pool() { static std::vector<char *> p; return p; }
    static std::mutex &pool_lock() { static std::mutex m; return m; }
    int take(size_t bytes, char **dev, char **host) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (off + bytes > CHUNK) {
            Chunk c{nullptr, nullptr};
            {
                std::lock_guard<std::mutex> g(pool_lock());
                if (!pool().empty()) { c.host = pool().back(); pool().pop_back(); }
            }
            if (!c.host && hipHostMalloc(reinterpret_cast<void **>(&c.host), CHUNK, hipHostMallocDefault) != hipSuccess) return 1;
            if (hipMalloc(reinterpret_cast<void **>(&c.dev), CHUNK) != hipSuccess) {
                std::lock_guard<std::mutex> g(pool_lock());
                pool().push_back(c.host);
                return 1;
            }
            chunks.push_back(c);
            off = 0;
        }
        *dev = chunks.back().dev + off;
        *host = chunks.back().host + off;
        off += bytes;
        return 0;
    }
    void release() {   // the engine's streams are idle
        std::lock_guard<std::mutex> g(pool_lock());
        for (auto &c : chunks) {
            (void)hipFree(c.dev);
            if (pool().size() < 8) pool().push_back(c.host);
            else (void)hipHostFree(c.host);
        }
        chunks.clear();
        off = CHUNK;
    }
};

struct Level {
    UploadArena *arena = nullptr;   // the engine's (set by mgrit_hip_create)
    WideHost *wide = nullptr;
    int order = 0;   // two-point steppers: BDF order (1 or 2)
    bool set = false;
    H2DHost *h2d = nullptr;
    LevelDev dev{};
    int G = 0, n_csets = 0, transfer = MGRIT_HIP_TRANSFER_COPY;
    std::vector<void *> allocs;
    std::vector<RunList> runs;
    std::vector<PairList> pairs;
    std::vector<IntervalsDev> ivals;   // mgrit_hip_intervals_create (device arrays live in allocs)
    std::vector<int> ivals_n;          // residual positions of the level (res_len) per list
    std::vector<int> ivals_cnt;        // intervals per list
    int same_factor_below = -1;        // the next coarser level has the same (one) forcing space factor, bit for bit; -1: not looked at yet
    std::vector<double> s_host;        // forcing space factors as uploaded (row storage order): levels with equal factors share
                                       // the LDS copy of the kernels that keep the factor there (FORCE 4)
    double *scratch = nullptr;
    size_t scratch_rows = 0;
    struct AtPlans { int32_t *d_src = nullptr, *d_own = nullptr; int count = 0; std::vector<std::vector<H2DPlan>> steps; };
    std::map<int, AtPlans> at_plans;   // batched truncated solves (mgrit_hip_at_solve on Heat2D / wide levels), by distance k
    double *gen_rows = nullptr;      // mgrit_hip_gen_down / _up: [res_len][ld] uncorrected chunk-end C-points
    size_t gen_len = 0;              // res_len they were sized for
    double *chain_state = nullptr;   // caller-owned [ld + CHAIN_STATE_TAIL]: carry-free part of the last point + carries
    bool chain_resume = false;
    bool chain_overlapped = true;    // mgrit_hip_chain_enable: the caller's (global) word on the overlapped chain
    double fac = 0.0;                // Heat1D: a / dx^2
    std::vector<double> t_host;      // the local time grid as described
    BlkDev blk{};                    // time-parallel forward solve (mgrit_hip_blk.inc); blk.r = 0: step by step
    std::shared_ptr<double> blk_q;   // the level's share of the sine-mode table (process-wide cache, blk_mode_table_cached)
    double *blk_part = nullptr;      // [G chunks][B][BLK_RMAX] the chunks' sums of the inner products (blk_project_kernel)
    int blk_state = -1;              // -1: not configured yet (mgrit_hip_block_solve_config), else configured
    double *blk_qt = nullptr;        // blk_one_kernel (small levels, one launch): the modes transposed; null: the six-launch form
    unsigned *blk_sync = nullptr;    //   its barrier counters (device) ...
    unsigned *blk_err = nullptr;     //   ... and the word a barrier that gave up sets (pinned, host-visible)
};

// ghost exchange (mgrit_hip_comm.inc): one direction of one pair of ranks
enum { LINK_NONE = 0, LINK_RCCL = 1, LINK_MAILBOX = 2 };
struct Mailbox { double *slots = nullptr; int n_slots = 0, slot_doubles = 0; };
struct Link {
    int kind = LINK_NONE;
    ncclComm_t comm = nullptr;   // LINK_RCCL
    int peer = 0;                // the peer's rank inside comm
    Mailbox *mb = nullptr;       // LINK_MAILBOX (caller-owned)
    uint64_t messages = 0, bytes = 0, messages_in = 0;
};

}  // namespace

struct mgrit_hip_engine;
namespace { void links_close(mgrit_hip_engine *e, bool abort); }

struct mgrit_hip_engine {
    int n_levels = 0;
    hipStream_t stream = nullptr;
    std::vector<Level> L;
    bool timing = false;
    struct TimeRec { int kind, lvl; hipEvent_t ev0, ev1; };
    std::vector<TimeRec> trecs;              // one per timed entry-point call since the last drain
    std::vector<hipEvent_t> ev_pool;         // events of drained records, reused
    hipEvent_t last0 = nullptr, last1 = nullptr;   // events of the most recent timed call (mgrit_hip_last_kernel_ms)
    int reserve = 0;              // mgrit_hip_set_reserve: CUs of XCD 0 the sweeps leave to the chain workers (0: program order)
    int *sched = nullptr;         // device counter block (1 KB, allocated with the first chain / planned launch): [0..2] {next, xcc0,
                                  // done} of the sweeps' item queue, [4..5] {tickets, done} of the chain's worker selection (both
                                  // reset themselves at the end of every launch), [8..9] {granule epoch base, workers done} of the
                                  // chain, [16..79] the dummy row of lane0_add, [96..159] per-CU claims of the chain's worker selection
    u64 *chain_gran = nullptr;    // [2][MAX_G][4] granules of the cross-workgroup chain
    unsigned *chain_err = nullptr;  // pinned, device-mapped: set by a worker whose bounded spin gave up
    double *pinned = nullptr;     // host staging buffer for small read-backs
    size_t pinned_len = 0;
    std::vector<double *> pinned_old;   // smaller buffers it has outgrown
    hipEvent_t ev_read = nullptr;
    double **mirror_cur = nullptr;   // device word: the slab that mirrors the corrected level-0 C-points of the running cycle
    int mirror_row0 = 0;
    std::vector<Link> links;      // mgrit_hip_link_*: the rank's ends of its exchange links
    double *xscratch = nullptr;   // exchange scratch (zeros of a fresh chain state / a dropped hand-over)
    size_t xscratch_len = 0;
    UploadArena arena;            // small uploads of every level (dev_upload)
};

namespace {

// Brackets one entry-point call with a pair of HIP events on the engine's stream when timing is on (mgrit_hip_set_timing):
// every sweep entry point carries one, so per-sweep device times can be read back with mgrit_hip_timing_drain.
struct Timed {
    mgrit_hip_engine *e;
    hipEvent_t a = nullptr, b = nullptr;
    int kind, lvl;
    static hipEvent_t get(mgrit_hip_engine *e) {
        hipEvent_t ev = nullptr;
        if (!e->ev_pool.empty()) { ev = e->ev_pool.back(); e->ev_pool.pop_back(); }
        else if (hipEventCreate(&ev) != hipSuccess) ev = nullptr;
        return ev;
    }
    Timed(mgrit_hip_engine *e_, int kind_, int lvl_) : e(e_), kind(kind_), lvl(lvl_) {
        if (!e || !e->timing) return;
        a = get(e); b = get(e);
        if (a && b) (void)hipEventRecord(a, e->stream);
    }
    ~Timed() {
        if (!a || !b) return;
        (void)hipEventRecord(b, e->stream);
        e->trecs.push_back({kind, lvl, a, b});
        e->last0 = a; e->last1 = b;
    }
};

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
// group-local backward scan of the power vector rho^(j'+1), j' < len (zero beyond): serial recurrences
void build_pt(const CSet &c, int len, double *pt) {
    std::vector<double> q(GROUP);
    double p = c.rho;
    for (int j = 0; j < GROUP; ++j) {
        q[j] = j < len ? p : 0.0;
        p = p * c.rho;
    }
    double z = q[GROUP - 1];
    pt[GROUP - 1] = z;
    for (int j = GROUP - 2; j >= 0; --j) {
        z = std::fma(c.rho, z, q[j]);
        pt[j] = z;
    }
}

// group-local image of src (len valid entries, zero beyond) under forward scan, zero padding, backward scan;
// serial recurrences. a = forward total, b = backward total. (The oracle's chain_local is the same text.)
void chain_local(const CSet &c, const double *src, int len, double *V, double *a, double *b) {
    std::vector<double> y(GROUP);
    y[0] = src[0];
    for (int j = 1; j < GROUP; ++j) y[j] = std::fma(c.rho, y[j - 1], src[j]);
    *a = y[GROUP - 1];
    for (int j = len; j < GROUP; ++j) y[j] = 0.0;
    double z = y[GROUP - 1];
    V[GROUP - 1] = z;
    for (int j = GROUP - 2; j >= 0; --j) {
        z = std::fma(c.rho, z, y[j]);
        V[j] = z;
    }
    *b = V[0];
}

// tables of the overlapped chain for one coefficient set, in the layout LevelDev::chT describes
void build_chain_tables(const CSet &c, const std::vector<double> &tab, const double *pt_full, const double *pt_last, int n,
                        int ld, std::vector<double> &out) {
    const int G = ld / GROUP, last_len = n - (G - 1) * GROUP;
    out.assign((size_t)ld + 6 * GROUP + 8 + 2 * MAX_G, 0.0);
    double *sc = out.data() + ld + 6 * GROUP;
    std::vector<double> src(GROUP), q1(GROUP), V(GROUP);
    for (int var = 0; var < 2; ++var) {
        const int len = var ? last_len : GROUP;
        const double *pt = var ? pt_last : pt_full;
        for (int j = 0; j < GROUP; ++j) q1[j] = j < len ? c.ik * pt[j] : 0.0;
        chain_local(c, q1.data(), len, V.data(), &sc[var], &sc[2 + var]);
        for (int j = 0; j < GROUP; ++j) {
            out[(size_t)ld + (size_t)var * GROUP + row_pos(j)] = q1[j];
            out[(size_t)ld + (size_t)(2 + var) * GROUP + row_pos(j)] = V[j];
        }
        for (int j = 0; j < GROUP; ++j) {
            const int l = j / E, k = j % E;
            src[j] = j < len ? (c.lp[LANES - 1 - l] * c.ik) * c.pw[E - k] : 0.0;
        }
        chain_local(c, src.data(), len, V.data(), &sc[4 + var], &sc[6 + var]);
        for (int j = 0; j < GROUP; ++j) out[(size_t)ld + (size_t)(4 + var) * GROUP + row_pos(j)] = V[j];
    }
    for (int g = 0; g < G; ++g) {
        const int len = g == G - 1 ? last_len : GROUP;
        for (int j = 0; j < GROUP; ++j) src[j] = j < len ? tab[(size_t)g * GROUP + j] : 0.0;
        chain_local(c, src.data(), len, V.data(), &sc[8 + g], &sc[8 + MAX_G + g]);
        for (int j = 0; j < GROUP; ++j) out[row_pos(g * GROUP + j)] = V[j];
    }
}

// rho^e by square-and-multiply, lowest bit first
double pow_int(double r, int e) {
    double res = 1.0, b = r;
    while (e > 0) {
        if (e & 1) res = res * b;
        e >>= 1;
        if (e) b = b * b;
    }
    return res;
}

// Rank-one correction table in closed form: w = A^{-1} e0 has the entries (rho^j - rho^(2n-j)) / (kappa (1 - rho^2)),
// so w-gamma_j = P_t * rho^k - Q_t * rho^(15-k) for element k of thread t = j / 16, with per-thread factors that are products of
// one per-group and one per-lane power. The sweep kernels evaluate exactly this expression (heat_w) instead of holding the
// n-entry table in LDS; kernels that keep a table (chain workers, two-point steppers) read the same bits from tabP.
void build_cset_heat1d(CSet &c, std::vector<double> &tab, int n, double fac, double dt) {
    const double beta = dt * fac;
    const double D = dt * (2.0 * fac) + 1.0;
    const double s = std::sqrt((D - 2.0 * beta) * (D + 2.0 * beta));
    const double kappa = 0.5 * (D + s);
    const double rho = beta / kappa;
    c.ik = 1.0 / kappa;
    c.scal = 0.0;
    cset_powers(c, rho);
    const double om = (1.0 - rho) * (1.0 + rho);
    const double w0 = ((1.0 - pow_int(rho, 2 * n)) * c.ik) / om;
    const double kr2 = beta * rho;
    const double gamma = kr2 / (1.0 + kr2 * w0);
    const double gp = (gamma * c.ik) / om;
    constexpr int GW = MGRIT_HIP_MAX_N_WIDE / GROUP;   // group factors for wide states too (their kernels read the table)
    double Gp[GW], pg[GW], qg[GW], qg2[GW];
    Gp[0] = 1.0;
    for (int g = 1; g < GW; ++g) Gp[g] = Gp[g - 1] * c.gc;
    const int t_last = (n - 1) / E, gL = t_last / LANES, lL = t_last % LANES, e0 = 2 * n - E * t_last - (E - 1);
    const double qb = gp * (e0 >= 0 ? pow_int(rho, e0) : (rho >= 1e-12 ? 1.0 / pow_int(rho, -e0) : 0.0));
    for (int g = 0; g < GW; ++g) {
        pg[g] = gp * Gp[g];
        qg[g] = g <= gL ? qb * Gp[gL - g] : 0.0;
        qg2[g] = g < gL ? qb * Gp[gL - g - 1] : 0.0;
    }
    for (int g = 0; g < MAX_G; ++g) { c.pg[g] = pg[g]; c.qg[g] = qg[g]; c.qg2[g] = qg2[g]; }
    tab.assign(n, 0.0);
    for (int j = 0; j < n; ++j) {
        const int t = j / E, k = j % E, g = t / LANES, l = t % LANES;
        const double P = pg[g] * c.lp[l];
        const double Q = (l <= lL ? qg[g] : qg2[g]) * c.lp[(lL - l) & (LANES - 1)];
        tab[j] = std::fma(-Q, c.pw[E - 1 - k], P * c.pw[k]);
    }
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
    c.pi_last = c.pw[(n - 1) % E + 1];   // the power at the last unknown's position inside its lane (the periodic closure reads it)
}

template <typename T>
int dev_upload(Level &lv, hipStream_t st, const std::vector<T> &h, T **out) {
    void *d = nullptr;
    const size_t bytes = sizeof(T) * (h.empty() ? 1 : h.size());
    if (lv.arena && bytes <= UploadArena::SMALL) {
        char *dv = nullptr, *hs = nullptr;
        if (lv.arena->take(bytes, &dv, &hs)) return fail(MGRIT_HIP_EHIP, "no memory for an upload of %zu bytes", bytes);
        if (!h.empty()) {
            std::memcpy(hs, h.data(), sizeof(T) * h.size());
            HIP_TRY(hipMemcpyAsync(dv, hs, sizeof(T) * h.size(), hipMemcpyHostToDevice, st));
        }
        *out = reinterpret_cast<T *>(dv);
        return 0;
    }
    HIP_TRY(hipMalloc(&d, bytes));
    lv.allocs.push_back(d);
    if (!h.empty()) {
        HIP_TRY(hipMemcpyAsync(d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    *out = static_cast<T *>(d);
    return 0;
}

// MGRIT_HIP_CHAIN_PLAIN=1: forward_solve with the plain per-step arithmetic of 3.3 everywhere (measurement switch)
bool plain_chain() {
    static const bool v = [] { const char *s = std::getenv("MGRIT_HIP_CHAIN_PLAIN"); return s && s[0] == '1'; }();
    return v;
}

// MGRIT_HIP_CHAIN_LOCAL_G: widest state (in groups of 1024 values) whose chain runs inside ONE workgroup (chain_local_kernel);
// default 4 = all that fit its LDS (measured per step, advection: 0.90 / 1.13 / 1.12 us at 2 / 3 / 4 groups against 1.37 / 1.42 /
// 1.44 us with one workgroup per group), 0 = never (measurement switch; the results do not depend on it)
int chain_local_max_g() {
    static const int v = [] {
        const char *s = std::getenv("MGRIT_HIP_CHAIN_LOCAL_G");
        const int g = s ? std::atoi(s) : CHAIN_LOCAL_MAX_G;
        return g < 0 ? 0 : g > CHAIN_LOCAL_MAX_G ? CHAIN_LOCAL_MAX_G : g;
    }();
    return v;
}

size_t smem_bytes(int G, int kind = MGRIT_HIP_STEPPER_HEAT1D) {
    return (size_t)(8 * G * LANES + (kind == MGRIT_HIP_STEPPER_ADVECTION1D ? 0 : 2 * 512)) * sizeof(double2) + (11 * MAX_G + 2 * LANES) * sizeof(double);
}

// time-parallel forward solve (mgrit_hip_blk.inc): the regular carve-up
size_t blk_smem_bytes(int G) { return smem_bytes(G); }

constexpr int MAX_G2 = MGRIT_HIP_MAX_N_2PTS / GROUP;  // two-point steppers: waves per half
size_t smem2_bytes(int G) { return (size_t)2 * (8 * G * LANES + 2 * 512) * sizeof(double2) + (12 * MAX_G + 2 * LANES) * sizeof(double); }

template <typename K>
int allow_big_lds(K kernel, size_t bytes = smem_bytes(MAX_G)) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

// Dispatch tables over the template space: kind (heat1d, advection1d) x forcing mode (0, 1, 2; advection has none).
#define FOR_EACH_STEPPER(X) X(MGRIT_HIP_STEPPER_HEAT1D, 0) X(MGRIT_HIP_STEPPER_HEAT1D, 1) X(MGRIT_HIP_STEPPER_HEAT1D, 2) \
    X(MGRIT_HIP_STEPPER_HEAT1D, 3) X(MGRIT_HIP_STEPPER_ADVECTION1D, 0)

// two-point steppers: BDF order (1, 2) x forcing mode (0, 1, 2)
#define FOR_EACH_2PTS(X) X(1, 0) X(1, 1) X(1, 2) X(2, 0) X(2, 1) X(2, 2)

bool g_attr_done = false;
int setup_kernel_attrs() {
    if (g_attr_done) return 0;
    int rc;
#define ATTR_RELAX(K, F, G_, R) if ((rc = allow_big_lds(relax_kernel<K, F, G_, R>))) return rc;
#define ATTR_ALL(K, F)                                                                                              \
    ATTR_RELAX(K, F, false, ROLE_F) ATTR_RELAX(K, F, true, ROLE_F) ATTR_RELAX(K, F, false, ROLE_C)                   \
    ATTR_RELAX(K, F, true, ROLE_C) ATTR_RELAX(K, F, false, ROLE_C_WEIGHTED) ATTR_RELAX(K, F, true, ROLE_C_WEIGHTED)  \
    ATTR_RELAX(K, F, true, ROLE_FC)                                                                                  \
    if ((rc = allow_big_lds(residual_kernel<K, F>))) return rc;                                                      \
    if ((rc = allow_big_lds(fas_fine_kernel<K, F>))) return rc;                                                      \
    if ((rc = allow_big_lds(fas_coarse_kernel<K, F>))) return rc;                                                    \
    if ((rc = allow_big_lds(fas_fused_kernel<K, F>))) return rc;                                                     \
    if ((rc = allow_big_lds(fas_fused_kernel<K, F, 512>))) return rc;                                                \
    if ((rc = allow_big_lds(ecf_kernel<K, F, false>))) return rc;                                                    \
    if ((rc = allow_big_lds(ecf_kernel<K, F, true>))) return rc;                                                     \
    if ((rc = allow_big_lds(at_kernel<K, F>))) return rc;                                                           \
    if ((rc = allow_big_lds(gen_down_kernel<K, F, false>))) return rc;                                               \
    if ((rc = allow_big_lds(gen_down_kernel<K, F, true>))) return rc;                                                \
    if ((rc = allow_big_lds(gen_up_kernel<K, F, false, false>))) return rc;                                          \
    if ((rc = allow_big_lds(gen_up_kernel<K, F, false, true>))) return rc;                                           \
    if ((rc = allow_big_lds(gen_up_kernel<K, F, true, false>))) return rc;                                           \
    if ((rc = allow_big_lds(gen_down_kernel<K, F, false, 512>))) return rc;                                          \
    if ((rc = allow_big_lds(gen_down_kernel<K, F, true, 512>))) return rc;                                           \
    if ((rc = allow_big_lds(gen_up_kernel<K, F, false, false, 512>))) return rc;                                     \
    if ((rc = allow_big_lds(gen_up_kernel<K, F, false, true, 512>))) return rc;                                      \
    if ((rc = allow_big_lds(gen_up_kernel<K, F, true, false, 512>))) return rc;
    FOR_EACH_STEPPER(ATTR_ALL)
#define ATTR_CHAIN_LOCAL(K, F)                                                                                       \
    if ((rc = allow_big_lds(chain_local_kernel<K, F, false>, chain_local_lds(CHAIN_LOCAL_MAX_G)))) return rc;          \
    if ((rc = allow_big_lds(chain_local_kernel<K, F, true>, chain_local_lds(CHAIN_LOCAL_MAX_G)))) return rc;           \
    if (K == MGRIT_HIP_STEPPER_ADVECTION1D) {                                                                          \
        if ((rc = allow_big_lds(chain_local_kernel<K, F, false, true>, chain_local_lds(CHAIN_LOCAL_MAX_G)))) return rc;  \
        if ((rc = allow_big_lds(chain_local_kernel<K, F, true, true>, chain_local_lds(CHAIN_LOCAL_MAX_G)))) return rc;   \
    }
    FOR_EACH_STEPPER(ATTR_CHAIN_LOCAL)
    if ((rc = allow_big_lds(cfas_kernel<0>))) return rc;
    if ((rc = allow_big_lds(cfas_kernel<2>))) return rc;
    if ((rc = allow_big_lds(ecfr_kernel<0, false, true>))) return rc;
    if ((rc = allow_big_lds(ecfr_kernel<2, false, true>))) return rc;
    if ((rc = allow_big_lds(ecfr_kernel<4, false, true>))) return rc;
    if ((rc = allow_big_lds(ecfr_kernel<0, true, false>))) return rc;
    if ((rc = allow_big_lds(ecfr_kernel<2, true, false>))) return rc;
    if ((rc = allow_big_lds(ecfr_kernel<4, true, false>))) return rc;
    if ((rc = allow_big_lds(fas_fused1_kernel<0, false>))) return rc;
    if ((rc = allow_big_lds(fas_fused1_kernel<2, false>))) return rc;
    if ((rc = allow_big_lds(fas_fused1_kernel<0, true>))) return rc;
    if ((rc = allow_big_lds(fas_fused1_kernel<2, true>))) return rc;
    if ((rc = allow_big_lds(fas_fused1_kernel<4, false>))) return rc;
    if ((rc = allow_big_lds(fas_fused1_kernel<4, true>))) return rc;
    if ((rc = allow_big_lds(jump_kernel))) return rc;
    // the sweeps' instances compiled for 512 threads (sweep_tb): up to 83 KB of LDS
#define ATTR_MID(F)                                                                                                                    \
    if ((rc = allow_big_lds(relax_kernel<MGRIT_HIP_STEPPER_HEAT1D, F, true, ROLE_FC, 512>))) return rc;                                \
    if ((rc = allow_big_lds(ecf_kernel<MGRIT_HIP_STEPPER_HEAT1D, F, true, 512>))) return rc;
    ATTR_MID(0) ATTR_MID(1) ATTR_MID(2)
#define ATTR_MID2(F)                                                                                                                   \
    if ((rc = allow_big_lds(ecfr_kernel<F, false, true, 512>))) return rc;                                                             \
    if ((rc = allow_big_lds(ecfr_kernel<F, true, false, 512>))) return rc;                                                             \
    if ((rc = allow_big_lds(fas_fused1_kernel<F, false, 512>))) return rc;                                                             \
    if ((rc = allow_big_lds(fas_fused1_kernel<F, true, 512>))) return rc;
    ATTR_MID2(0) ATTR_MID2(2) ATTR_MID2(4)
    if ((rc = allow_big_lds(cfas_kernel<0, 512>))) return rc;
    if ((rc = allow_big_lds(cfas_kernel<2, 512>))) return rc;
#define ATTR_BLK(F)                                                                                                   \
    if ((rc = allow_big_lds(blk_local_kernel<MGRIT_HIP_STEPPER_HEAT1D, F>, blk_smem_bytes(MAX_G)))) return rc;        \
    if ((rc = allow_big_lds(blk_finish_kernel<MGRIT_HIP_STEPPER_HEAT1D, F>, blk_smem_bytes(MAX_G)))) return rc;
    ATTR_BLK(0) ATTR_BLK(2) ATTR_BLK(3) ATTR_BLK(4)
    if ((rc = allow_big_lds(blk_local_kernel<MGRIT_HIP_STEPPER_ADVECTION1D, 0>, blk_smem_bytes(MAX_G)))) return rc;
    if ((rc = allow_big_lds(blk_finish_kernel<MGRIT_HIP_STEPPER_ADVECTION1D, 0>, blk_smem_bytes(MAX_G)))) return rc;
    if ((rc = allow_big_lds(adv_fft_rows_kernel, (size_t)BLK_FOURIER_MAX_N * sizeof(double2)))) return rc;
    if ((rc = allow_big_lds(adv_dft_fwd_kernel, (size_t)BLK_FOURIER_MAX_N * sizeof(double2)))) return rc;
    if ((rc = allow_big_lds(adv_dft_inv_kernel, (size_t)BLK_FOURIER_MAX_N * sizeof(double2)))) return rc;
#define ATTR_2PTS(O, F)                                                                                              \
    if ((rc = allow_big_lds(relax2_kernel<O, F, false, false>, smem2_bytes(MAX_G2)))) return rc;                     \
    if ((rc = allow_big_lds(relax2_kernel<O, F, true, false>, smem2_bytes(MAX_G2)))) return rc;                      \
    if ((rc = allow_big_lds(relax2_kernel<O, F, false, true>, smem2_bytes(MAX_G2)))) return rc;                      \
    if ((rc = allow_big_lds(relax2_kernel<O, F, true, true>, smem2_bytes(MAX_G2)))) return rc;                       \
    if ((rc = allow_big_lds(residual2_kernel<O, F>, smem2_bytes(MAX_G2)))) return rc;                                \
    if ((rc = allow_big_lds(fas_fine2_kernel<O, F>, smem2_bytes(MAX_G2)))) return rc;                                \
    if ((rc = allow_big_lds(fas_coarse2_kernel<O, F>, smem2_bytes(MAX_G2)))) return rc;                              \
    if ((rc = allow_big_lds(at2_kernel<O, F>, smem2_bytes(MAX_G2)))) return rc;
    FOR_EACH_2PTS(ATTR_2PTS)
    if ((rc = allow_big_lds(jump2_kernel, smem2_bytes(MAX_G2)))) return rc;
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
    const int n_max = MGRIT_HIP_MAX_N_WIDE;
    if (n < 1 || n > n_max)
        return fail(MGRIT_HIP_EUNSUPPORTED, "n=%d DOFs per time point outside [1,%d] (register-resident up to 16384, three-launch Phi above)", n, n_max);
    if (ld != mgrit_hip_row_stride(n)) return fail(MGRIT_HIP_EINVAL, "ld=%d must equal mgrit_hip_row_stride(n=%d)=%d", ld, n, mgrit_hip_row_stride(n));
    if (n_pts < 0 || (n_pts > 0 && !t_local)) return fail(MGRIT_HIP_EINVAL, "bad local time grid");
    if (K < 0 || K > 8 || (K > 0 && (!s || (n_pts > 0 && !tau)))) return fail(MGRIT_HIP_EINVAL, "bad forcing description (K=%d)", K);
    if ((rc = setup_kernel_attrs())) return rc;
    Level &lv = e->L[lvl];
    if (lv.set) return fail(MGRIT_HIP_EINVAL, "level %d already described", lvl);
    const int G = (n + GROUP - 1) / GROUP, T = G * LANES;
    lv.G = G;
    lv.fac = fac;
    if (n_pts > 0) lv.t_host.assign(t_local, t_local + n_pts);
    LevelDev &d = lv.dev;
    d.n = n; d.ld = ld; d.T = T; d.n_pts = n_pts; d.K = K; d.kind = kind;
    d.stream_rows = (size_t)n_pts * (size_t)ld * sizeof(double) > ((size_t)256 << 20) ? 1 : 0;
    // coefficient sets keyed by the bit pattern of dt = t[i] - t[i-1] (the reference uses each step's own dt)
    std::vector<double> dts(n_pts > 0 ? n_pts : 0, 0.0), uniq;
    std::vector<int32_t> cidx(n_pts > 0 ? n_pts : 0, 0);
    std::map<uint64_t, int> seen;   // bit pattern of dt -> coefficient set
    for (int i = 1; i < n_pts; ++i) {
        const double dt = t_local[i] - t_local[i - 1];
        dts[i] = dt;
        uint64_t bits;
        std::memcpy(&bits, &dt, sizeof(double));
        auto it = seen.find(bits);
        if (it == seen.end()) {
            // every distinct step size costs one set of tables (8*ld + 16 KB each, resident in HBM: 9.4 GB of the 288 for a
            // fully non-uniform grid of 65536 steps at n = 16382) and a table reload per step
            if (uniq.size() >= (1u << 18))
                return fail(MGRIT_HIP_EUNSUPPORTED, "more than 262144 distinct time-step sizes on level %d (one coefficient table per size)", lvl);
            it = seen.emplace(bits, (int)uniq.size()).first;
            uniq.push_back(dt);
        }
        cidx[i] = it->second;
    }
    if (n_pts > 0) cidx[0] = 0;
    lv.n_csets = (int)uniq.size();
    d.one_cset = uniq.size() == 1 ? 1 : 0;
    std::vector<CSet> cs(uniq.size());
    std::vector<double> tabT(uniq.size() * (size_t)E * T, 0.0), tab;  // row storage order per coefficient set
    std::vector<double> ptT(uniq.size() * (size_t)2 * GROUP, 0.0), pt(GROUP), pt_last(GROUP), chT;
    // the overlapped chain (the oracle's chain_overlapped is the same rule on its one rank)
    const bool overlapped = kind == MGRIT_HIP_STEPPER_HEAT1D && G >= 2 && G <= MAX_G && uniq.size() == 1 && K <= 1;
    if (n > MGRIT_HIP_MAX_N) lv.wide = new WideHost();
    for (size_t q = 0; q < uniq.size(); ++q) {
        std::memset(&cs[q], 0, sizeof(CSet));
        if (kind == MGRIT_HIP_STEPPER_HEAT1D) {
            build_cset_heat1d(cs[q], tab, n, fac, uniq[q]);
            build_pt(cs[q], GROUP, pt.data());
            cs[q].pi_full = pt[0];
            for (int j = 0; j < GROUP; ++j) ptT[q * (size_t)2 * GROUP + row_pos(j)] = pt[j];
            build_pt(cs[q], n - ((n - 1) / GROUP) * GROUP, pt_last.data());
            cs[q].pi_last = pt_last[0];
            for (int j = 0; j < GROUP; ++j) ptT[q * (size_t)2 * GROUP + GROUP + row_pos(j)] = pt_last[j];
            if (overlapped) build_chain_tables(cs[q], tab, pt.data(), pt_last.data(), n, ld, chT);
        } else build_cset_advection1d(cs[q], tab, n, fac, uniq[q]);
        for (int j = 0; j < n; ++j) tabT[q * (size_t)E * T + row_pos(j)] = tab[j];
    }
    std::vector<double> sT((size_t)(K > 0 ? K : 0) * E * T, 0.0), tauv;
    for (int kk = 0; kk < K; ++kk)
        for (int j = 0; j < n; ++j) sT[(size_t)kk * E * T + row_pos(j)] = s[(size_t)kk * n + j];
    if (K > 0) {  // tc[k][i] = tau_k(t_i) * dt_i
        tauv.assign(tau, tau + (size_t)K * n_pts);
        for (int kk = 0; kk < K; ++kk)
            for (int i = 0; i < n_pts; ++i) tauv[(size_t)kk * n_pts + i] = tauv[(size_t)kk * n_pts + i] * dts[i];
    }
    int32_t *d_cidx; double *d_dt, *d_tau, *d_sT, *d_tabT, *d_ptT; CSet *d_cs;
    if ((rc = dev_upload(lv, e->stream, cidx, &d_cidx))) return rc;
    if ((rc = dev_upload(lv, e->stream, dts, &d_dt))) return rc;
    if ((rc = dev_upload(lv, e->stream, tauv, &d_tau))) return rc;
    lv.s_host = sT;
    if ((rc = dev_upload(lv, e->stream, sT, &d_sT))) return rc;
    if ((rc = dev_upload(lv, e->stream, cs, &d_cs))) return rc;
    if ((rc = dev_upload(lv, e->stream, tabT, &d_tabT))) return rc;
    if ((rc = dev_upload(lv, e->stream, ptT, &d_ptT))) return rc;
    d.cidx = d_cidx; d.dt = d_dt; d.tc = d_tau; d.cs = d_cs;
    d.ptP = reinterpret_cast<const double2 *>(d_ptT);
    d.sP = reinterpret_cast<const double2 *>(d_sT);
    d.tabP = reinterpret_cast<const double2 *>(d_tabT);
    d.chT = nullptr;
    if (overlapped) {
        double *d_chT;
        if ((rc = dev_upload(lv, e->stream, chT, &d_chT))) return rc;
        d.chT = d_chT;
    }
    lv.set = true;
    return 0;
}

// BDF2 coefficients of heat_1d_2pts_bdf2.py:103-110 for the pair of step sizes (tau_i, tau_im1), rewritten for the scaled
// system (I + L/c) x = rhs/c (the oracle's bdf2_half is the same text)
struct HalfCoef { double dt_eff, a, nb, fs; };
HalfCoef bdf2_half(double tau_i, double tau_im1) {
    HalfCoef h;
    const double r = tau_i / tau_im1;
    const double cm2 = (r * r) / (tau_i * (1.0 + r));
    const double cm1 = (1.0 + r) / tau_i;
    const double c = (1.0 + 2.0 * r) / (tau_i * (1.0 + r));
    const double inv = 1.0 / c;
    h.dt_eff = inv; h.a = cm1 * inv; h.nb = -(cm2 * inv); h.fs = inv;
    return h;
}

int level_heat1d_2pts(mgrit_hip_engine *e, int lvl, int n_pts, const double *t_local, int n, int ld, double fac, double dtau,
                      int order, int K, const double *s, const double *tau, const double *tau2) {
    int rc = check_level(e, lvl, false);
    if (rc) return rc;
    if (n < 1 || n > MGRIT_HIP_MAX_N_WIDE)
        return fail(MGRIT_HIP_EUNSUPPORTED, "n=%d DOFs per time point outside [1,%d] (two-point stepper)", n, MGRIT_HIP_MAX_N_WIDE);
    if (ld != 2 * mgrit_hip_row_stride(n)) return fail(MGRIT_HIP_EINVAL, "ld=%d must equal 2*mgrit_hip_row_stride(n=%d)=%d", ld, n, 2 * mgrit_hip_row_stride(n));
    if (order != 1 && order != 2) return fail(MGRIT_HIP_EINVAL, "BDF order must be 1 or 2");
    if (n_pts < 0 || (n_pts > 0 && !t_local)) return fail(MGRIT_HIP_EINVAL, "bad local time grid");
    if (K < 0 || K > 8 || (K > 0 && (!s || (n_pts > 0 && (!tau || !tau2))))) return fail(MGRIT_HIP_EINVAL, "bad forcing description (K=%d)", K);
    if ((rc = setup_kernel_attrs())) return rc;
    Level &lv = e->L[lvl];
    if (lv.set) return fail(MGRIT_HIP_EINVAL, "level %d already described", lvl);
    const int G = (n + GROUP - 1) / GROUP, T = G * LANES;
    lv.G = G;
    lv.order = order;
    if (n > MGRIT_HIP_MAX_N_2PTS) lv.wide = new WideHost();   // wider than a workgroup holds: every half-solve as three launches (mgrit_hip_wide.inc)
    LevelDev &d = lv.dev;
    d.n = n; d.ld = ld; d.T = T; d.n_pts = n_pts; d.K = K; d.kind = MGRIT_HIP_STEPPER_HEAT1D_2PTS;
    d.stream_rows = 0;
    const size_t np = n_pts > 0 ? n_pts : 0;
    std::vector<double> uniq, hc(np * 4, 0.0), fs(np * 2, 0.0), dts(np, 0.0);
    std::vector<int32_t> cidx2(np * 2, 0);
    auto set_of = [&](double dt_eff) -> int {
        for (size_t q = 0; q < uniq.size(); ++q)
            if (std::memcmp(&uniq[q], &dt_eff, sizeof(double)) == 0) return (int)q;
        uniq.push_back(dt_eff);
        return (int)uniq.size() - 1;
    };
    for (int i = 1; i < n_pts; ++i) {
        const double t_start = t_local[i - 1], t_stop = t_local[i];
        dts[i] = t_stop - t_start;
        const double tl0 = t_stop - t_start - dtau;
        HalfCoef h[2];
        if (order == 1) {
            h[0].dt_eff = tl0; h[1].dt_eff = dtau;
            for (int q = 0; q < 2; ++q) { h[q].a = 1.0; h[q].nb = 0.0; h[q].fs = h[q].dt_eff; }
        } else {
            h[0] = bdf2_half(tl0, dtau);
            h[1] = bdf2_half(dtau, tl0);
        }
        for (int q = 0; q < 2; ++q) {
            cidx2[2 * (size_t)i + q] = set_of(h[q].dt_eff);
            hc[4 * (size_t)i + 2 * q] = h[q].a;
            hc[4 * (size_t)i + 2 * q + 1] = h[q].nb;
            fs[2 * (size_t)i + q] = h[q].fs;
        }
        if (uniq.size() > 4096) return fail(MGRIT_HIP_EUNSUPPORTED, "more than 4096 distinct time-step sizes on level %d", lvl);
    }
    lv.n_csets = (int)uniq.size();
    std::vector<CSet> cs(uniq.size());
    std::vector<double> tabT(uniq.size() * (size_t)E * T, 0.0), tab;
    std::vector<double> ptT(uniq.size() * (size_t)2 * GROUP, 0.0), pt(GROUP);
    for (size_t q = 0; q < uniq.size(); ++q) {
        std::memset(&cs[q], 0, sizeof(CSet));
        build_cset_heat1d(cs[q], tab, n, fac, uniq[q]);
        build_pt(cs[q], GROUP, pt.data());
        cs[q].pi_full = pt[0];
        for (int j = 0; j < GROUP; ++j) ptT[q * (size_t)2 * GROUP + row_pos(j)] = pt[j];
        build_pt(cs[q], n - ((n - 1) / GROUP) * GROUP, pt.data());
        cs[q].pi_last = pt[0];
        for (int j = 0; j < GROUP; ++j) ptT[q * (size_t)2 * GROUP + GROUP + row_pos(j)] = pt[j];
        for (int j = 0; j < n; ++j) tabT[q * (size_t)E * T + row_pos(j)] = tab[j];
    }
    std::vector<double> sT((size_t)(K > 0 ? K : 0) * E * T, 0.0), tc((size_t)(K > 0 ? K : 0) * np * 2, 0.0);
    for (int kk = 0; kk < K; ++kk) {
        for (int j = 0; j < n; ++j) sT[(size_t)kk * E * T + row_pos(j)] = s[(size_t)kk * n + j];
        for (int i = 1; i < n_pts; ++i) {   // tc[k][i][half] = tau_k(t_i [+ dtau]) * forcing scale of the half
            tc[((size_t)kk * np + i) * 2] = tau[(size_t)kk * n_pts + i] * fs[2 * (size_t)i];
            tc[((size_t)kk * np + i) * 2 + 1] = tau2[(size_t)kk * n_pts + i] * fs[2 * (size_t)i + 1];
        }
    }
    int32_t *d_cidx2; double *d_dt, *d_tc, *d_sT, *d_tabT, *d_ptT, *d_hc; CSet *d_cs;
    if ((rc = dev_upload(lv, e->stream, cidx2, &d_cidx2))) return rc;
    if ((rc = dev_upload(lv, e->stream, dts, &d_dt))) return rc;
    if ((rc = dev_upload(lv, e->stream, tc, &d_tc))) return rc;
    if ((rc = dev_upload(lv, e->stream, sT, &d_sT))) return rc;
    if ((rc = dev_upload(lv, e->stream, cs, &d_cs))) return rc;
    if ((rc = dev_upload(lv, e->stream, tabT, &d_tabT))) return rc;
    if ((rc = dev_upload(lv, e->stream, ptT, &d_ptT))) return rc;
    if ((rc = dev_upload(lv, e->stream, hc, &d_hc))) return rc;
    d.cidx = nullptr; d.cidx2 = d_cidx2; d.hc = d_hc; d.dt = d_dt; d.tc = d_tc; d.cs = d_cs;
    d.ptP = reinterpret_cast<const double2 *>(d_ptT);
    d.sP = reinterpret_cast<const double2 *>(d_sT);
    d.tabP = reinterpret_cast<const double2 *>(d_tabT);
    lv.set = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Heat2D host side: tables, batch plans, the batched Phi pipeline
// ---------------------------------------------------------------------------------------------------------------
constexpr int H2D_MAX_BATCH = 1024;  // items per GEMM batch (work buffers: 2 x 1024 x Mi x Mj doubles; 2048 measured no faster)

uint64_t dbl_bits(double v) { uint64_t b; std::memcpy(&b, &v, 8); return b; }

template <typename T>
int dev_upload_raw(Level &lv, hipStream_t st, const T *h, size_t n, T **out) {
    std::vector<T> tmp(h, h + n);
    return dev_upload(lv, st, tmp, out);
}

// group batch items by the time-step size of their step index (one D table per group)
struct H2DItem { int32_t in, step, dst, a, b; };

int h2d_make_plans(mgrit_hip_engine *e, Level &lv, const std::vector<H2DItem> &items, std::vector<H2DPlan> &plans) {
    H2DHost &h = *lv.h2d;
    std::vector<uint64_t> keys;
    for (const H2DItem &it : items) {
        const uint64_t k = dbl_bits(h.dts[it.step]);
        bool seen = false;
        for (uint64_t q : keys) seen = seen || q == k;
        if (!seen) keys.push_back(k);
    }
    for (uint64_t k : keys) {
        std::vector<int32_t> vin, vst, vds, va, vb;
        for (const H2DItem &it : items)
            if (dbl_bits(h.dts[it.step]) == k) {
                vin.push_back(it.in); vst.push_back(it.step); vds.push_back(it.dst); va.push_back(it.a); vb.push_back(it.b);
            }
        for (size_t off = 0; off < vin.size(); off += H2D_MAX_BATCH) {
            const size_t cnt = std::min<size_t>(H2D_MAX_BATCH, vin.size() - off);
            H2DPlan pl;
            pl.count = (int)cnt;
            pl.dtbits = k;
            int rc;
            if ((rc = dev_upload_raw(lv, e->stream, vin.data() + off, cnt, &pl.d_in))) return rc;
            if ((rc = dev_upload_raw(lv, e->stream, vst.data() + off, cnt, &pl.d_step))) return rc;
            if ((rc = dev_upload_raw(lv, e->stream, vds.data() + off, cnt, &pl.d_dst))) return rc;
            if ((rc = dev_upload_raw(lv, e->stream, va.data() + off, cnt, &pl.d_a))) return rc;
            if ((rc = dev_upload_raw(lv, e->stream, vb.data() + off, cnt, &pl.d_b))) return rc;
            plans.push_back(pl);
        }
    }
    return 0;
}

int h2d_reserve(Level &lv, int count) {
    H2DHost &h = *lv.h2d;
    if ((size_t)count <= h.cap_items) return 0;
    if (h.W0) HIP_TRY(hipFree(h.W0));
    if (h.W1) HIP_TRY(hipFree(h.W1));
    if (h.rowsq) HIP_TRY(hipFree(h.rowsq));
    h.W0 = h.W1 = h.rowsq = nullptr;
    const size_t per = (size_t)h.dev.Mi * h.dev.Mj;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h.W0), sizeof(double) * per * count));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h.W1), sizeof(double) * per * count));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h.rowsq), sizeof(double) * (size_t)h.dev.nx * count));
    h.cap_items = count;
    return 0;
}

int h2d_dinv(mgrit_hip_engine *e, Level &lv, uint64_t dtbits, double **out) {
    H2DHost &h = *lv.h2d;
    auto it = h.dinv.find(dtbits);
    if (it != h.dinv.end()) { *out = it->second; return 0; }
    double dt;
    std::memcpy(&dt, &dtbits, 8);
    const double thdt = h.dev.theta * dt;
    std::vector<double> tab((size_t)h.dev.Mi * h.dev.Mj, 0.0);   // spectral slot order on both axes, 0 where no mode lives
    const int hxe = (h.dev.mi + 1) / 2, hxo = h.dev.mi / 2, hye = (h.dev.mj + 1) / 2, hyo = h.dev.mj / 2;
    for (int a = 0; a < h.dev.Mi; ++a) {
        if (!((a < hxe) || (a >= h.HPx && a < h.HPx + hxo))) continue;
        for (int b = 0; b < h.dev.Mj; ++b)
            if ((b < hye) || (b >= h.HPy && b < h.HPy + hyo))
                tab[(size_t)a * h.dev.Mj + b] = 1.0 / (1.0 + thdt * (h.lx[a] + h.ly[b]));
    }
    double *d = nullptr;
    int rc = dev_upload(lv, e->stream, tab, &d);
    if (rc) return rc;
    h.dinv[dtbits] = d;
    *out = d;
    return 0;
}

// U (in W0) = interior of Phi applied to the rows in_slab[plan.d_in[b]] for the steps plan.d_step[b]  (theta > 0)
// fin: the sweep's arithmetic is applied by the last transform itself (h2d_inv_kernel<true> + h2d_rim_kernel): no epilogue
// launch, and the interior of U is neither written to nor read from the work slab
int h2d_phi_batch(mgrit_hip_engine *e, Level &lv, const H2DPlan &pl, const double *in_slab, bool chain = false,
                  const H2DFin *fin = nullptr) {
    H2DHost &h = *lv.h2d;
    const H2DDev &H = h.dev;
    if (H.theta == 0.0) return 0;  // explicit: evaluated inside the epilogue kernels
    int rc;
    const size_t per = (size_t)H.Mi * H.Mj;
    chain = chain && pl.count == 1;
    if (chain && !h.Wc0) {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h.Wc0), sizeof(double) * per));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h.Wc1), sizeof(double) * per));
    }
    if (!chain && (rc = h2d_reserve(lv, std::min(H2D_MAX_BATCH, std::max(pl.count, 1))))) return rc;
    double *const W0 = chain ? h.Wc0 : h.W0, *const W1 = chain ? h.Wc1 : h.W1;
    double *dinv = nullptr;
    if ((rc = h2d_dinv(e, lv, pl.dtbits, &dinv))) return rc;
    // W1[j][i'] = x to spectral slots ; W0[i'][j'] = y to spectral slots, o D ; W1[j'][i] = x back ; W0[i][j] = y back = U
    const dim3 fx(H.Mj / 64, H.Mi / 64, pl.count), fy(H.Mi / 64, H.Mj / 64, pl.count);        // (n tiles, slot tiles, items)
    const dim3 ix(H.Mj / 64, h.HPx / 64, pl.count), iy(H.Mi / 64, h.HPy / 64, pl.count);      // (n tiles, i tiles, items)
    if (H.theta == 1.0 && H.K == 0 && !H.fb && !H.has_w && H.mj >= 2) {
        // the homogeneous backward-Euler step: the right-hand side IS the interior of u -- the first transform stages it from the
        // state rows (h2d_kloop<.., GRID>), no rhs launch
        hipLaunchKernelGGL((h2d_fwd_kernel<false, true>), fx, dim3(256), 0, e->stream, h.Fxe, h.Fxo, H.mi, h.HPx, in_slab, H.ny, W1,
                           nullptr, per, pl.d_in, H.ld, H.mj);
    } else {
        hipLaunchKernelGGL(h2d_rhs_kernel, dim3((H.Mj + 255) / 256, H.Mi, pl.count), dim3(256), 0, e->stream, H, in_slab, pl.d_in,
                           pl.d_step, W0);
        hipLaunchKernelGGL((h2d_fwd_kernel<false>), fx, dim3(256), 0, e->stream, h.Fxe, h.Fxo, H.mi, h.HPx, W0, H.Mj, W1, nullptr, per);
    }
    hipLaunchKernelGGL((h2d_fwd_kernel<true>), fy, dim3(256), 0, e->stream, h.Fye, h.Fyo, H.mj, h.HPy, W1, H.Mi, W0, dinv, per);
    const H2DFin none{};
    hipLaunchKernelGGL((h2d_inv_kernel<false>), ix, dim3(256), 0, e->stream, h.FxeT, h.FxoT, H.mi, h.HPx, W0, H.Mj, W1, per, H, none);
    if (fin) {
        hipLaunchKernelGGL((h2d_inv_kernel<true>), iy, dim3(256), 0, e->stream, h.FyeT, h.FyoT, H.mj, h.HPy, W1, H.Mi, W0, per, H, *fin);
        const int n_rim = 2 * H.ny + 2 * (H.nx - 2);
        hipLaunchKernelGGL(h2d_rim_kernel, dim3((n_rim + 127) / 128, pl.count), dim3(128), 0, e->stream, H, *fin);
    } else {
        hipLaunchKernelGGL((h2d_inv_kernel<false>), iy, dim3(256), 0, e->stream, h.FyeT, h.FyoT, H.mj, h.HPy, W1, H.Mi, W0, per, H, none);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// Phi of the batch followed by the sweep's arithmetic: fused into the last transform for the implicit schemes, the epilogue
// kernel (which evaluates the explicit stencil itself) for theta = 0.
int h2d_finish(mgrit_hip_engine *e, Level &lv, const H2DPlan &pl, const double *in_slab, double *dst_slab, int dst_ld,
               const double *a_slab, const double *b_slab, int op, int use_g, double w, bool chain);

int h2d_phi_op(mgrit_hip_engine *e, Level &lv, const H2DPlan &pl, const double *in_slab, double *dst_slab, int dst_ld,
               const double *a_slab, const double *b_slab, int op, int use_g, double w, bool chain = false) {
    int rc;
    // (the sequential coarsest-level solve keeps the epilogue kernel: fused, one step of one state took 107 instead of 92 us --
    // the last transform's 64 tiles then also carry the sweep's loads and the rim is a launch of its own)
    if (lv.h2d->dev.theta != 0.0 && !chain) {
        const H2DFin fin{dst_slab, dst_ld, pl.d_dst, a_slab, pl.d_a, b_slab, pl.d_b, op, use_g, w, 1.0 - w};
        return h2d_phi_batch(e, lv, pl, in_slab, chain, &fin);
    }
    if ((rc = h2d_phi_batch(e, lv, pl, in_slab, chain))) return rc;
    return h2d_finish(e, lv, pl, in_slab, dst_slab, dst_ld, a_slab, b_slab, op, use_g, w, chain);
}

int h2d_finish(mgrit_hip_engine *e, Level &lv, const H2DPlan &pl, const double *in_slab, double *dst_slab, int dst_ld,
               const double *a_slab, const double *b_slab, int op, int use_g, double w, bool chain) {
    const H2DDev &H = lv.h2d->dev;
    hipLaunchKernelGGL(h2d_finish_kernel, dim3((H.ny + 127) / 128, H.nx, pl.count), dim3(128), 0, e->stream, H,
                       (chain && pl.count == 1 && lv.h2d->Wc0) ? lv.h2d->Wc0 : lv.h2d->W0, in_slab,
                       pl.d_in, pl.d_step, dst_slab, dst_ld, pl.d_dst, a_slab, pl.d_a, b_slab, pl.d_b, op, use_g, w, 1.0 - w);
    HIP_TRY(hipGetLastError());
    return 0;
}

int h2d_relax(mgrit_hip_engine *e, int lvl, RunList *rl, int mode, double weight_c) {
    Level &lv = e->L[lvl];
    int rc;
    if (!rl->h2d_relax_built) {
        int maxlen = 0;
        for (int r = 0; r < rl->n; ++r) maxlen = std::max(maxlen, (int)rl->h_len[r]);
        for (int k = 0; k < maxlen; ++k) {  // step k of every run that is long enough: one batch
            std::vector<H2DItem> items;
            for (int r = 0; r < rl->n; ++r)
                if (rl->h_len[r] > k) {
                    const int i = rl->h_start[r] + k;
                    items.push_back({i - 1, i, i, i, i});
                }
            std::vector<H2DPlan> plans;
            if ((rc = h2d_make_plans(e, lv, items, plans))) return rc;
            for (H2DPlan &p : plans) rl->h2d_relax.push_back(p);
        }
        rl->h2d_relax_built = true;
    }
    const int op = mode == MGRIT_HIP_RELAX_C ? H2D_OP_C : H2D_OP_F;
    const bool chain = mode == MGRIT_HIP_RELAX_CHAIN && lv.h2d->dev.theta != 0.0;   // its own work buffers (H2DHost::Wc0)
    for (const H2DPlan &pl : rl->h2d_relax) {
        if ((rc = h2d_phi_op(e, lv, pl, lv.dev.u, lv.dev.u, lv.dev.ld, lv.dev.g, lv.dev.u, op, lvl > 0 ? 1 : 0, weight_c, chain))) return rc;
    }
    return 0;
}

// Time-parallel forward solve of a Heat2D level with backward Euler (the oracle's heat2d_block_solve_spec): the level's
// steps in blocks of BLK_K. First pass: step s of EVERY block as one batch -- the error x_b a block has accumulated,
// x_b = (g_i + Phi(u_{i-1} + x_b)) - u_i from x_b = 0; the errors at the block ends through the forward transforms, the recurrence
// over the blocks on the full spectrum, the propagated parts back through the inverse transforms straight into the rows
// (h2d_inv_kernel<true>, H2D_OP_ADD); second pass: the block interiors from the corrected block starts, again one batch per step
// index. ~2 K batches of B states instead of K B single-state steps.
bool h2d_block_ok(const Level &lv, int lvl) {
    // backward Euler, and (round 5) Crank-Nicolson: the explicit half of a step is diagonal in the same sine basis for an error
    // whose rim is zero -- every state of the level carries the boundary values there --, which h2d_block_solve checks per solve
    if (!lv.h2d || lvl == 0 || !(lv.h2d->dev.theta == 1.0 || lv.h2d->dev.theta == 0.5)) return false;
    const int N = lv.dev.n_pts - 1;
    return N >= 4 * MGRIT_HIP_BLOCK_K && N / MGRIT_HIP_BLOCK_K <= H2D_MAX_BATCH;
}

int h2d_block_build(mgrit_hip_engine *e, Level &lv) {
    H2DHost &h = *lv.h2d;
    H2DHost::Blk &k = h.blk;
    const H2DDev &H = h.dev;
    const int K = MGRIT_HIP_BLOCK_K, N = lv.dev.n_pts - 1, B = N / K;
    const size_t per = (size_t)H.Mi * H.Mj;
    int rc;
    auto first = [&](int b) { return K * b + 1; };
    auto last = [&](int b) { return b == B - 1 ? N : K * (b + 1); };
    int maxlen = 0;
    for (int b = 0; b < B; ++b) maxlen = std::max(maxlen, last(b) - first(b) + 1);
    k.p1.resize(maxlen); k.p3.resize(maxlen);
    for (int s = 0; s < maxlen; ++s) {
        std::vector<H2DItem> it1, it3;
        for (int b = 0; b < B; ++b) {
            const int i = first(b) + s;
            if (i > last(b)) continue;
            // first pass: input row = u_{i-1} (s = 0) or Y[b] = u_{i-1} + x_b; output x_b = (g_i + Phi) - u_i into X[b]
            it1.push_back({s == 0 ? i - 1 : b, i, b, i, i});
            if (i < last(b)) it3.push_back({i - 1, i, i, i, i});
        }
        if (!it1.empty() && (rc = h2d_make_plans(e, lv, it1, k.p1[s]))) return rc;
        if (!it3.empty() && (rc = h2d_make_plans(e, lv, it3, k.p3[s]))) return rc;
    }
    std::vector<int32_t> iota(B), end_all, end_tail, dsel(B, 0);
    for (int b = 0; b < B; ++b) { iota[b] = b; end_all.push_back(last(b)); }
    for (int b = 1; b < B; ++b) end_tail.push_back(last(b));
    // propagator tables: the elementwise product of the steps' D tables in step order; one table per distinct sequence of step sizes
    std::map<std::vector<uint64_t>, int> seen;
    std::vector<double> tabs, dv(per);
    const int hxe = (H.mi + 1) / 2, hxo = H.mi / 2, hye = (H.mj + 1) / 2, hyo = H.mj / 2;
    for (int b = 1; b < B; ++b) {
        std::vector<uint64_t> key;
        for (int i = first(b); i <= last(b); ++i) key.push_back(dbl_bits(h.dts[i]));
        auto it = seen.find(key);
        if (it == seen.end()) {
            const size_t off = tabs.size();
            tabs.resize(off + per, 0.0);
            for (int i = first(b); i <= last(b); ++i) {
                const double thdt = H.theta * h.dts[i];
                std::fill(dv.begin(), dv.end(), 0.0);
                for (int a = 0; a < H.Mi; ++a) {
                    if (!((a < hxe) || (a >= h.HPx && a < h.HPx + hxo))) continue;
                    for (int c = 0; c < H.Mj; ++c)
                        if ((c < hye) || (c >= h.HPy && c < h.HPy + hyo)) {
                            const double lam = h.lx[a] + h.ly[c], inv = 1.0 / (1.0 + thdt * lam);
                            // (theta < 1: the step's explicit half carries theta as well, heat_2d.py:309; oracle h2d_prop_table)
                            dv[(size_t)a * H.Mj + c] = H.theta == 1.0 ? inv : (1.0 - thdt * lam) * inv;
                        }
                }
                if (i == first(b)) std::copy(dv.begin(), dv.end(), tabs.begin() + (long)off);
                else for (size_t q = 0; q < per; ++q) tabs[off + q] = tabs[off + q] * dv[q];
            }
            it = seen.emplace(key, (int)seen.size()).first;
        }
        dsel[b] = it->second;
    }
    if ((rc = dev_upload(lv, e->stream, iota, &k.d_iota))) return rc;
    if ((rc = dev_upload(lv, e->stream, end_all, &k.d_end_all))) return rc;
    if ((rc = dev_upload(lv, e->stream, end_tail, &k.d_end_tail))) return rc;
    if ((rc = dev_upload(lv, e->stream, dsel, &k.d_dsel))) return rc;
    if ((rc = dev_upload(lv, e->stream, tabs, &k.Dtab))) return rc;
    for (double **buf : {&k.spec, &k.X, &k.Y}) {
        const size_t doubles = (buf == &k.spec ? per : (size_t)lv.dev.ld) * (size_t)B;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(buf), sizeof(double) * doubles));
        lv.allocs.push_back(*buf);
        HIP_TRY(hipMemsetAsync(*buf, 0, sizeof(double) * doubles, e->stream));
    }
    if ((rc = h2d_reserve(lv, std::min(H2D_MAX_BATCH, B)))) return rc;
    if (H.theta != 1.0) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&k.rim_flag), 256, hipHostMallocMapped));
        *k.rim_flag = 0u;
        HIP_TRY(hipEventCreateWithFlags(&k.rim_ev, hipEventDisableTiming));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    k.B = B;
    k.built = true;
    return 0;
}

// *stepped = true: the level has to be solved step by step after all (theta < 1 and an error with a non-zero rim); nothing was changed
int h2d_block_solve(mgrit_hip_engine *e, int lvl, bool *stepped) {
    *stepped = false;
    Level &lv = e->L[lvl];
    H2DHost &h = *lv.h2d;
    H2DHost::Blk &k = h.blk;
    const H2DDev &H = h.dev;
    int rc;
    if (!k.built && (rc = h2d_block_build(e, lv))) return rc;
    const int B = k.B, ld = lv.dev.ld, n = H.nx * H.ny;
    const size_t per = (size_t)H.Mi * H.Mj;
    // first pass: x_b = (g_i + Phi(u_{i-1} + x_b)) - u_i, step s of every block as one batch
    for (size_t s = 0; s < k.p1.size(); ++s)
        for (const H2DPlan &pl : k.p1[s]) {
            if (s > 0)   // Y[b] = u_{i-1} + X[b]  (d_in = b, d_a = i)
                hipLaunchKernelGGL(h2d_sum_rows_kernel, dim3((n + 255) / 256, pl.count), dim3(256), 0, e->stream, ld, n, k.Y, pl.d_in,
                                   lv.dev.u, pl.d_a, -1, k.X, pl.d_in);
            if ((rc = h2d_phi_op(e, lv, pl, s == 0 ? lv.dev.u : k.Y, k.X, ld, lv.dev.g, lv.dev.u, H2D_OP_DEFECT, 1, 1.0))) return rc;
        }
    if (H.theta != 1.0) {
        // the explicit half of a step reads the rim of its input: the modes describe the propagation of an error whose rim is
        // zero. That holds whenever every state of the level carries the boundary values on its rim (any state a Phi has produced
        // does); a level that still holds an initial guess with another rim -- C-points never relaxed, cf_iter = 0 -- is stepped.
        // One word read back per solve (a Heat2D forward solve is tens of milliseconds of launches).
        *k.rim_flag = 0u;
        hipLaunchKernelGGL(h2d_rim_check_kernel, dim3(B - 1), dim3(256), 0, e->stream, H, k.X, ld, k.rim_flag);
        HIP_TRY(hipEventRecord(k.rim_ev, e->stream));
        for (;;) {
            const hipError_t q = hipEventQuery(k.rim_ev);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return fail(MGRIT_HIP_EHIP, "hipEventQuery: %s", hipGetErrorString(q));
        }
        if (*k.rim_flag != 0u) { *stepped = true; return 0; }
    }
    // spectra of the errors at the block ends e_0 .. e_{B-2}
    const dim3 fx(H.Mj / 64, H.Mi / 64, B - 1), fy(H.Mi / 64, H.Mj / 64, B - 1);
    hipLaunchKernelGGL(h2d_pack_kernel, dim3((H.Mj + 255) / 256, H.Mi, B - 1), dim3(256), 0, e->stream, H, k.X, k.d_iota, h.W0);
    hipLaunchKernelGGL((h2d_fwd_kernel<false>), fx, dim3(256), 0, e->stream, h.Fxe, h.Fxo, H.mi, h.HPx, h.W0, H.Mj, h.W1, nullptr, per);
    hipLaunchKernelGGL((h2d_fwd_kernel<false>), fy, dim3(256), 0, e->stream, h.Fye, h.Fyo, H.mj, h.HPy, h.W1, H.Mi, k.spec, nullptr, per);
    // recurrence over the blocks; block ends: u[e_b] = (u[e_b] + x_b) + the propagated part on the interior (b >= 1)
    hipLaunchKernelGGL(h2d_blk_scan_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, e->stream, k.spec, k.Dtab, k.d_dsel, B, per);
    hipLaunchKernelGGL(h2d_sum_rows_kernel, dim3((n + 255) / 256, B), dim3(256), 0, e->stream, ld, n, lv.dev.u, k.d_end_all, lv.dev.u,
                       k.d_end_all, 0, k.X, k.d_iota);
    const dim3 ix(H.Mj / 64, h.HPx / 64, B - 1), iy(H.Mi / 64, h.HPy / 64, B - 1);
    const H2DFin none{};
    const H2DFin add{lv.dev.u, ld, k.d_end_tail, nullptr, nullptr, nullptr, nullptr, H2D_OP_ADD, 0, 1.0, 0.0};
    hipLaunchKernelGGL((h2d_inv_kernel<false>), ix, dim3(256), 0, e->stream, h.FxeT, h.FxoT, H.mi, h.HPx, k.spec + per, H.Mj, h.W1, per, H, none);
    hipLaunchKernelGGL((h2d_inv_kernel<true>), iy, dim3(256), 0, e->stream, h.FyeT, h.FyoT, H.mj, h.HPy, h.W1, H.Mi, h.W0, per, H, add);
    HIP_TRY(hipGetLastError());
    // second pass: every block's interior from its (corrected) start
    for (size_t s = 0; s < k.p3.size(); ++s)
        for (const H2DPlan &pl : k.p3[s])
            if ((rc = h2d_phi_op(e, lv, pl, lv.dev.u, lv.dev.u, ld, lv.dev.g, lv.dev.u, H2D_OP_F, 1, 1.0))) return rc;
    return 0;
}

// per-point sums of squares: Phi(u_{i-1}) - u_i (residual) or u_i - prev_i (jump)
int h2d_points_sumsq(mgrit_hip_engine *e, int lvl, RunList *rl, const double *prev, double *out) {
    Level &lv = e->L[lvl];
    const H2DDev &H = lv.h2d->dev;
    int rc;
    if (!rl->h2d_points_built) {
        std::vector<H2DItem> items;
        for (int r = 0; r < rl->n; ++r) {
            const int i = rl->h_start[r];
            items.push_back({i - 1, i, i, i, r});   // (b: the point's position in the list = where its sum goes)
        }
        if ((rc = h2d_make_plans(e, lv, items, rl->h2d_points))) return rc;
        rl->h2d_points_built = true;
    }
    for (const H2DPlan &pl : rl->h2d_points) {
        if ((rc = h2d_reserve(lv, std::min(H2D_MAX_BATCH, pl.count)))) return rc;
        if (!prev) {
            if ((rc = h2d_phi_batch(e, lv, pl, lv.dev.u))) return rc;
            hipLaunchKernelGGL(h2d_rowsq_kernel, dim3((H.nx + 63) / 64, pl.count), dim3(64), 0, e->stream, H, lv.h2d->W0, lv.dev.u,
                               pl.d_in, pl.d_step, lv.dev.u, pl.d_dst, H2D_OP_RESIDUAL, lv.h2d->rowsq);
        } else {
            hipLaunchKernelGGL(h2d_rowsq_kernel, dim3((H.nx + 63) / 64, pl.count), dim3(64), 0, e->stream, H, lv.h2d->W0, lv.dev.u,
                               pl.d_dst, pl.d_step, prev, pl.d_dst, H2D_OP_JUMP, lv.h2d->rowsq);
        }
        hipLaunchKernelGGL(h2d_rowsum_kernel, dim3((pl.count + 63) / 64), dim3(64), 0, e->stream, lv.h2d->rowsq, H.nx, pl.count,
                           out, pl.d_b);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

int h2d_fas_rhs(mgrit_hip_engine *e, int lvl, PairList *pl) {
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    int rc;
    if (!pl->h2d_built) {
        std::vector<H2DItem> fine, coarse;
        for (int p = 0; p < pl->n; ++p) {
            const int i = pl->h_fine[p], j = pl->h_coarse[p];
            fine.push_back({i - 1, i, j, i, i});      // dst = g^{l+1}_j, a = g^l_i, b = u^l_i
            coarse.push_back({j - 1, j, j, j, j});    // in = v_{j-1}, dst = g_j, a = g_j, b = v_j
        }
        if ((rc = h2d_make_plans(e, lf, fine, pl->h2d_fine))) return rc;
        if ((rc = h2d_make_plans(e, lc, coarse, pl->h2d_coarse))) return rc;
        pl->h2d_built = true;
    }
    for (const H2DPlan &q : pl->h2d_fine) {
        if ((rc = h2d_phi_op(e, lf, q, lf.dev.u, lc.dev.g, lc.dev.ld, lf.dev.g, lf.dev.u, H2D_OP_FAS_FINE, lvl > 0 ? 1 : 0, 1.0))) return rc;
    }
    for (const H2DPlan &q : pl->h2d_coarse) {
        if ((rc = h2d_phi_op(e, lc, q, lc.dev.v, lc.dev.g, lc.dev.ld, lc.dev.g, lc.dev.v, H2D_OP_FAS_COARSE, 1, 1.0))) return rc;
    }
    return 0;
}

// folded sine tables of one axis (the oracle's h2d_axis_tables is the same text): Fe[k][e] = Q[k][2e],
// Fo[k][o] = Q[k][2o+1] for k < ceil(m/2) resp. floor(m/2), their transposes, and the eigenvalues in spectral slot order
void h2d_axis_tables(int m, int HP, double f, std::vector<double> &Fe, std::vector<double> &Fo, std::vector<double> &FeT,
                     std::vector<double> &FoT, std::vector<double> &lam) {
    const int hE = (m + 1) / 2, hO = m / 2;
    const double sc = std::sqrt(2.0 / (m + 1));
    Fe.assign((size_t)HP * HP, 0.0); Fo.assign((size_t)HP * HP, 0.0);
    FeT.assign((size_t)HP * HP, 0.0); FoT.assign((size_t)HP * HP, 0.0);
    lam.assign((size_t)2 * HP, 0.0);
    for (int k = 0; k < hE; ++k) {
        for (int e = 0; e < hE; ++e) {
            const long r = ((long)(k + 1) * (2 * e + 1)) % (2L * (m + 1));
            Fe[(size_t)k * HP + e] = FeT[(size_t)e * HP + k] = sc * std::sin(M_PI * (double)r / (double)(m + 1));
        }
        if (k < hO)
            for (int o = 0; o < hO; ++o) {
                const long r = ((long)(k + 1) * (2 * o + 2)) % (2L * (m + 1));
                Fo[(size_t)k * HP + o] = FoT[(size_t)o * HP + k] = sc * std::sin(M_PI * (double)r / (double)(m + 1));
            }
    }
    for (int e = 0; e < hE; ++e) { const double hs = std::sin(M_PI * (double)(2 * e + 1) / (2.0 * (m + 1))); lam[e] = 4.0 * f * hs * hs; }
    for (int o = 0; o < hO; ++o) { const double hs = std::sin(M_PI * (double)(2 * o + 2) / (2.0 * (m + 1))); lam[HP + o] = 4.0 * f * hs * hs; }
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
    const int n_pts = lv.dev.n_pts;
    if (n_pts == 0) return 0;   // a rank that owns no point of this level has nothing to bind
    if (!lv.dev.u) return fail(MGRIT_HIP_EINVAL, "state slabs not bound");
    if (need_vg && (!lv.dev.v || !lv.dev.g)) return fail(MGRIT_HIP_EINVAL, "v/g slabs not bound");
    return 0;
}

// grid of persistent workgroups: as many as stay resident on the chip (LDS- and thread-limited), at most one per item
bool is_2pts(const Level &lv) { return lv.dev.kind == MGRIT_HIP_STEPPER_HEAT1D_2PTS; }

// MGRIT_HIP_GEN_512=0: the general whole-level passes in their 1024-thread instances everywhere (measurement and comparison; same bits)
bool gen_half_instances() {
    const char *s = std::getenv("MGRIT_HIP_GEN_512");
    return !(s && s[0] == '0' && s[1] == 0);
}

int wgs_per_cu(const Level &lv) {
    const size_t lds = is_2pts(lv) ? smem2_bytes(lv.G) : smem_bytes(lv.G, lv.dev.kind);
    return std::max(1, std::min((int)(160 * 1024 / lds), 2048 / lv.dev.T));
}

int persistent_grid(const Level &lv, int n_items) { return std::min(n_items, 256 * wgs_per_cu(lv)); }

int force_mode(const Level &lv) {
    if (lv.dev.kind != MGRIT_HIP_STEPPER_HEAT1D && lv.dev.kind != MGRIT_HIP_STEPPER_HEAT1D_2PTS) return 0;
    if (lv.dev.fb) return 3;
    return lv.dev.K == 0 ? 0 : lv.dev.K == 1 ? 1 : 2;
}


// ---------------------------------------------------------------------------------------------------------------
// Wide Heat1D states (mgrit_hip_wide.inc): batch plans and the three-launch Phi, in the scheme of the Heat2D path
// ---------------------------------------------------------------------------------------------------------------
constexpr int WIDE_MAX_BATCH = 2048;   // items per batch (work slab: 2048 rows of up to 512 KB)

int wide_reserve(Level &lv, int count) {
    WideHost &h = *lv.wide;
    if ((size_t)count <= h.cap) return 0;
    for (double **p : {&h.W, &h.tot, &h.car, &h.z0, &h.red, &h.T0}) {
        if (*p) lv.allocs.push_back(*p);   // kept until the engine goes: a captured cycle may still launch with the old addresses
        *p = nullptr;
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h.W), sizeof(double) * (size_t)count * lv.dev.ld));
    if (lv.dev.kind == MGRIT_HIP_STEPPER_HEAT1D_2PTS)
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h.T0), sizeof(double) * (size_t)count * (lv.dev.ld / 2)));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h.tot), sizeof(double) * (size_t)count * 2 * WIDE_MAX_G));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h.car), sizeof(double) * (size_t)count * 2 * WIDE_MAX_G));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h.z0), sizeof(double) * (size_t)count));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h.red), sizeof(double) * (size_t)count * 2 * WIDE_MAX_G));   // (two-point: both halves)
    h.cap = count;
    return 0;
}

int wide_make_plans(mgrit_hip_engine *e, Level &lv, const std::vector<H2DItem> &items, std::vector<H2DPlan> &plans) {
    for (size_t off = 0; off < items.size(); off += WIDE_MAX_BATCH) {
        const size_t cnt = std::min<size_t>(WIDE_MAX_BATCH, items.size() - off);
        std::vector<int32_t> vin(cnt), vst(cnt), vds(cnt), va(cnt), vb(cnt);
        for (size_t k = 0; k < cnt; ++k) {
            const H2DItem &it = items[off + k];
            vin[k] = it.in; vst[k] = it.step; vds[k] = it.dst; va[k] = it.a; vb[k] = it.b;
        }
        H2DPlan pl;
        pl.count = (int)cnt;
        int rc;
        if ((rc = dev_upload_raw(lv, e->stream, vin.data(), cnt, &pl.d_in))) return rc;
        if ((rc = dev_upload_raw(lv, e->stream, vst.data(), cnt, &pl.d_step))) return rc;
        if ((rc = dev_upload_raw(lv, e->stream, vds.data(), cnt, &pl.d_dst))) return rc;
        if ((rc = dev_upload_raw(lv, e->stream, va.data(), cnt, &pl.d_a))) return rc;
        if ((rc = dev_upload_raw(lv, e->stream, vb.data(), cnt, &pl.d_b))) return rc;
        plans.push_back(pl);
    }
    return 0;
}

// the scanned rows of Phi(in_slab[plan.d_in[b]]) for the steps plan.d_step[b] in the work slab, and their carries
int wide_phi(mgrit_hip_engine *e, Level &lv, const H2DPlan &pl, const double *in_slab) {
    int rc;
    if ((rc = wide_reserve(lv, std::min(WIDE_MAX_BATCH, std::max(pl.count, 1))))) return rc;
    WideHost &h = *lv.wide;
    const int G = lv.dev.T / LANES;
    const dim3 grid((G + 15) / 16, pl.count), block(1024);
    const int fm = force_mode(lv);
    if (lv.dev.kind == MGRIT_HIP_STEPPER_HEAT1D_2PTS) {   // two half-solves; the second one's scanned rows and carries are what wide_finish takes
        for (int hf = 0; hf < 2; ++hf) {
#define WIDE2_LOCAL(O_, F_)                                                                                                         \
    if (lv.order == O_ && (fm != 0) == (F_ != 0))                                                                                  \
        hipLaunchKernelGGL((wide2_local_kernel<O_, F_>), grid, block, 0, e->stream, lv.dev, hf, in_slab, pl.d_in, pl.d_step, h.T0, h.W, h.tot);
            WIDE2_LOCAL(1, 0) WIDE2_LOCAL(1, 2) WIDE2_LOCAL(2, 0) WIDE2_LOCAL(2, 2)
            hipLaunchKernelGGL(wide_carry_kernel, dim3(pl.count), dim3(64), 0, e->stream, lv.dev, pl.d_step, h.tot, h.car, h.z0, hf);
            if (hf == 0)
                hipLaunchKernelGGL(wide2_finish_kernel, grid, block, 0, e->stream, lv.dev, 0, h.W, pl.d_step, h.car, h.z0, h.T0, (double *)nullptr, 0,
                                   (const int32_t *)nullptr, (const double *)nullptr, (const int32_t *)nullptr, (const double *)nullptr,
                                   (const int32_t *)nullptr, 0, 0, 1.0, 0.0, (double *)nullptr);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (fm == 0) hipLaunchKernelGGL((wide_local_kernel<0>), grid, block, 0, e->stream, lv.dev, in_slab, pl.d_in, pl.d_step, h.W, h.tot);
    else if (fm == 3) hipLaunchKernelGGL((wide_local_kernel<3>), grid, block, 0, e->stream, lv.dev, in_slab, pl.d_in, pl.d_step, h.W, h.tot);
    else hipLaunchKernelGGL((wide_local_kernel<2>), grid, block, 0, e->stream, lv.dev, in_slab, pl.d_in, pl.d_step, h.W, h.tot);
    hipLaunchKernelGGL(wide_carry_kernel, dim3(pl.count), dim3(64), 0, e->stream, lv.dev, pl.d_step, h.tot, h.car, h.z0, -1);
    HIP_TRY(hipGetLastError());
    return 0;
}

int wide_finish(mgrit_hip_engine *e, Level &lv, const H2DPlan &pl, double *dst_slab, int dst_ld, const double *a_slab,
                const double *b_slab, int op, int use_g, double w) {
    WideHost &h = *lv.wide;
    const int G = lv.dev.T / LANES;
    if (lv.dev.kind == MGRIT_HIP_STEPPER_HEAT1D_2PTS)
        hipLaunchKernelGGL(wide2_finish_kernel, dim3((G + 15) / 16, pl.count), dim3(1024), 0, e->stream, lv.dev, 1, h.W, pl.d_step, h.car, h.z0,
                           h.T0, dst_slab, dst_ld, pl.d_dst, a_slab, pl.d_a, b_slab, pl.d_b, op, use_g, w, 1.0 - w, h.red);
    else
    hipLaunchKernelGGL(wide_finish_kernel, dim3((G + 15) / 16, pl.count), dim3(1024), 0, e->stream, lv.dev, h.W, pl.d_step, h.car, h.z0,
                       dst_slab, dst_ld, pl.d_dst, a_slab, pl.d_a, b_slab, pl.d_b, op, use_g, w, 1.0 - w, h.red);
    HIP_TRY(hipGetLastError());
    return 0;
}

int wide_relax(mgrit_hip_engine *e, int lvl, RunList *rl, int mode, double weight_c) {
    Level &lv = e->L[lvl];
    int rc;
    if (mode == MGRIT_HIP_RELAX_FC) return fail(MGRIT_HIP_EUNSUPPORTED, "relax mode FC: states of at most %d values", MGRIT_HIP_MAX_N);
    if (!rl->h2d_relax_built) {
        int maxlen = 0;
        for (int r = 0; r < rl->n; ++r) maxlen = std::max(maxlen, (int)rl->h_len[r]);
        for (int k = 0; k < maxlen; ++k) {   // step k of every run that is long enough: one batch
            std::vector<H2DItem> items;
            for (int r = 0; r < rl->n; ++r)
                if (rl->h_len[r] > k) {
                    const int i = rl->h_start[r] + k;
                    items.push_back({i - 1, i, i, i, i});
                }
            if ((rc = wide_make_plans(e, lv, items, rl->h2d_relax))) return rc;
        }
        rl->h2d_relax_built = true;
    }
    const int op = mode == MGRIT_HIP_RELAX_C ? WIDE_OP_C : WIDE_OP_F;
    for (const H2DPlan &pl : rl->h2d_relax) {
        if ((rc = wide_phi(e, lv, pl, lv.dev.u))) return rc;
        if ((rc = wide_finish(e, lv, pl, lv.dev.u, lv.dev.ld, lv.dev.g, lv.dev.u, op, lvl > 0 ? 1 : 0, weight_c))) return rc;
    }
    return 0;
}

int wide_points_sumsq(mgrit_hip_engine *e, int lvl, RunList *rl, const double *prev, double *out) {
    Level &lv = e->L[lvl];
    int rc;
    if (!rl->h2d_points_built) {
        std::vector<H2DItem> items;
        for (int r = 0; r < rl->n; ++r) {
            const int i = rl->h_start[r];
            items.push_back({i - 1, i, i, i, i});
        }
        if ((rc = wide_make_plans(e, lv, items, rl->h2d_points))) return rc;
        rl->h2d_points_built = true;
    }
    const int G = lv.dev.T / LANES, halves = lv.dev.kind == MGRIT_HIP_STEPPER_HEAT1D_2PTS ? 2 : 1;
    int off = 0;
    for (const H2DPlan &pl : rl->h2d_points) {
        if ((rc = wide_reserve(lv, std::min(WIDE_MAX_BATCH, std::max(pl.count, 1))))) return rc;
        if (!prev) {
            if ((rc = wide_phi(e, lv, pl, lv.dev.u))) return rc;
            if ((rc = wide_finish(e, lv, pl, lv.dev.u, lv.dev.ld, lv.dev.u, lv.dev.u, WIDE_OP_RESIDUAL, 0, 1.0))) return rc;
        } else {
            hipLaunchKernelGGL(wide_diffsq_kernel, dim3((G + 15) / 16, pl.count), dim3(1024), 0, e->stream, lv.dev, lv.dev.u, prev,
                               pl.d_dst, lv.wide->red, halves);
        }
        hipLaunchKernelGGL(wide_rowsum_kernel, dim3((pl.count + 63) / 64), dim3(64), 0, e->stream, lv.wide->red, halves * G, pl.count, out + off,
                           halves * WIDE_MAX_G);
        HIP_TRY(hipGetLastError());
        off += pl.count;
    }
    return 0;
}

// fine half of the FAS right-hand side for a wide fine level: rows into dst_slab (g^{l+1} itself for the copy transfer, the
// level's scratch rows -- one per pair -- for a spatial transfer that follows)
int wide_fas_fine(mgrit_hip_engine *e, int lvl, PairList *pl, double *dst_slab, int dst_ld, bool by_pair) {
    Level &lf = e->L[lvl];
    int rc;
    if (pl->h2d_fine.empty()) {
        std::vector<H2DItem> fine;
        for (int p = 0; p < pl->n; ++p) {
            const int i = pl->h_fine[p];
            fine.push_back({i - 1, i, by_pair ? p : pl->h_coarse[p], i, i});
        }
        if ((rc = wide_make_plans(e, lf, fine, pl->h2d_fine))) return rc;
    }
    for (const H2DPlan &q : pl->h2d_fine) {
        if ((rc = wide_phi(e, lf, q, lf.dev.u))) return rc;
        if ((rc = wide_finish(e, lf, q, dst_slab, dst_ld, lf.dev.g, lf.dev.u, WIDE_OP_FAS_FINE, lvl > 0 ? 1 : 0, 1.0))) return rc;
    }
    return 0;
}

int wide_fas_coarse(mgrit_hip_engine *e, int lvl, PairList *pl) {
    Level &lc = e->L[lvl + 1];
    int rc;
    if (pl->h2d_coarse.empty()) {
        std::vector<H2DItem> coarse;
        for (int p = 0; p < pl->n; ++p) {
            const int j = pl->h_coarse[p];
            coarse.push_back({j - 1, j, j, j, j});
        }
        if ((rc = wide_make_plans(e, lc, coarse, pl->h2d_coarse))) return rc;
    }
    for (const H2DPlan &q : pl->h2d_coarse) {
        if ((rc = wide_phi(e, lc, q, lc.dev.v))) return rc;
        if ((rc = wide_finish(e, lc, q, lc.dev.g, lc.dev.ld, lc.dev.g, lc.dev.v, WIDE_OP_FAS_COARSE, 1, 1.0))) return rc;
    }
    return 0;
}

// entry points that hold a state in one workgroup refuse wide levels
int no_wide(const Level &a, const Level *b, const char *what) {
    if (a.wide || (b && b->wide))
        return fail(MGRIT_HIP_EUNSUPPORTED, "%s: states of at most %d values per time point (wider Heat1D states run sweep by sweep)", what, MGRIT_HIP_MAX_N);
    return 0;
}

// two-point kernels: template space BDF order x forcing mode
#define LAUNCH2_CASE(kernel, O_, F_, lv, grid, ...)                                                              \
    if ((lv).order == O_ && force_mode(lv) == F_)                                                                 \
        hipLaunchKernelGGL((kernel<O_, F_>), dim3(grid), dim3((lv).dev.T), smem2_bytes((lv).G), e->stream, __VA_ARGS__);
#define LAUNCH2_BY_ORDER(kernel, lv, grid, ...)                                                                  \
    do {                                                                                                         \
        LAUNCH2_CASE(kernel, 1, 0, lv, grid, __VA_ARGS__) LAUNCH2_CASE(kernel, 1, 1, lv, grid, __VA_ARGS__)       \
        LAUNCH2_CASE(kernel, 1, 2, lv, grid, __VA_ARGS__) LAUNCH2_CASE(kernel, 2, 0, lv, grid, __VA_ARGS__)       \
        LAUNCH2_CASE(kernel, 2, 1, lv, grid, __VA_ARGS__) LAUNCH2_CASE(kernel, 2, 2, lv, grid, __VA_ARGS__)       \
        HIP_TRY(hipGetLastError());                                                                              \
    } while (0)

#define LAUNCH_CASE(kernel, K_, F_, lv, grid, ...)                                                               \
    if ((lv).dev.kind == K_ && force_mode(lv) == F_)                                                              \
        hipLaunchKernelGGL((kernel<K_, F_>), dim3(grid), dim3((lv).dev.T), smem_bytes((lv).G, (lv).dev.kind), e->stream, __VA_ARGS__);
#define LAUNCH_BY_KIND(kernel, lv, grid, ...)                                                                    \
    do {                                                                                                         \
        LAUNCH_CASE(kernel, MGRIT_HIP_STEPPER_HEAT1D, 0, lv, grid, __VA_ARGS__)                                   \
        LAUNCH_CASE(kernel, MGRIT_HIP_STEPPER_HEAT1D, 1, lv, grid, __VA_ARGS__)                                   \
        LAUNCH_CASE(kernel, MGRIT_HIP_STEPPER_HEAT1D, 2, lv, grid, __VA_ARGS__)                                   \
        LAUNCH_CASE(kernel, MGRIT_HIP_STEPPER_HEAT1D, 3, lv, grid, __VA_ARGS__)                                   \
        LAUNCH_CASE(kernel, MGRIT_HIP_STEPPER_ADVECTION1D, 0, lv, grid, __VA_ARGS__)                              \
        HIP_TRY(hipGetLastError());                                                                              \
    } while (0)

int ensure_sched(mgrit_hip_engine *e) {
    if (e->sched) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(e->stream, &cs);
    if (cs != hipStreamCaptureStatusNone) return fail(MGRIT_HIP_EINVAL, "first chain / planned launch inside a stream capture");
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->sched), 1024));
    HIP_TRY(hipMemsetAsync(e->sched, 0, 1024, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return 0;
}

// level description as a kernel argument, with the launch-time scheduling fields filled in (see WgQueue): only the kernels
// that walk their items through a WgQueue look at them
LevelDev sched_dev(const mgrit_hip_engine *e, const Level &lv) {
    LevelDev d = lv.dev;
    d.sched = e->reserve > 0 ? e->sched : nullptr;
    d.xcc0_limit = e->reserve > 0 ? (32 - e->reserve) * wgs_per_cu(lv) : -1;
    return d;
}

// ---------------------------------------------------------------------------------------------------------------
// Time-parallel forward solve (mgrit_hip_blk.inc): the rule and the tables. The oracle's
// orc_block_solve_rank / heat1d_block_solve_spec state the same arithmetic independently.
// ---------------------------------------------------------------------------------------------------------------
constexpr double BLK_THR = 8.673617379884035e-19;   // 2^-60

double blk_lam4(int n, int k) {
    const double h = std::sin(M_PI * (double)(k + 1) / (2.0 * (double)(n + 1)));
    return 4.0 * h * h;
}
int blk_count(int nt) { const int N = nt - 1; return N >= 4 * BLK_K ? N / BLK_K : 0; }

// D[b][k], k <= BLK_RMAX (row stride BLK_RMAX + 1), and the rank; 0: not eligible
int blk_rank(int n, double fac, int nt, const double *t, std::vector<double> *D) {
    const int B = blk_count(nt);
    int r = 0;
    if (B == 0 || n < 2) return 0;
    const int KM = BLK_RMAX + 1 < n ? BLK_RMAX + 1 : n;
    if (D) D->assign((size_t)B * (BLK_RMAX + 1), 0.0);
    for (int b = 0; b < B; ++b) {
        const int first = BLK_K * b + 1, last = b == B - 1 ? nt - 1 : BLK_K * (b + 1);
        int rb = -1;
        for (int k = 0; k < KM; ++k) {
            const double lam = blk_lam4(n, k);
            double d = 1.0;
            for (int i = first; i <= last; ++i) d = d * (1.0 / (1.0 + ((t[i] - t[i - 1]) * fac) * lam));
            if (D) (*D)[(size_t)b * (BLK_RMAX + 1) + k] = d;
            if (rb < 0 && d < BLK_THR) rb = k;
        }
        if (b >= 1) {
            if (rb < 0 || rb > BLK_RMAX) return 0;
            if (rb > r) r = rb;
        }
    }
    return r < 1 ? 1 : r;
}

// cos and sin of one angle as two separate libm calls (a compiler that merges them into sincos() gets another last bit for a
// few arguments on glibc; the oracle's tables are built the same way)
double __attribute__((noinline)) sep_cos(double x) { return std::cos(x); }
double __attribute__((noinline)) sep_sin(double x) { return std::sin(x); }

int blk_wgs_per_cu(const Level &lv) { return std::max(1, std::min((int)(160 * 1024 / blk_smem_bytes(lv.G)), 2048 / lv.dev.T)); }

// doubles behind the point in the hand-over of a sharded solve: the amplitudes at the rank's last point
int blk_handover_len(const Level &lv) { return lv.blk.r == 0 ? 0 : lv.blk.fourier ? 2 * lv.dev.n : BLK_RMAX; }

// Advection1D: the Fourier form for 64 <= n <= BLK_FOURIER_MAX_N -- n = 2^p: radix-2 transforms inside one workgroup's LDS (a row's n
// complex values); any other n: the transforms as ordered sums on the matrix cores (adv_dft_*_kernel)
bool blk_fourier_ok(int n, int nt) { return n >= 64 && n <= BLK_FOURIER_MAX_N && blk_count(nt) > 0; }

// MGRIT_HIP_BLK_ONE=0: small levels take the six launches too (measurement and comparison switch; same bits)
bool blk_one_launch_enabled() {
    const char *s = std::getenv("MGRIT_HIP_BLK_ONE");
    return !(s && std::atoi(s) == 0);
}

int blk_launch(mgrit_hip_engine *e, Level &lv, int phases) {
    BlkDev &bk = lv.blk;
    const int fm = force_mode(lv), F = fm == 1 ? 4 : fm;
    const size_t lds = blk_smem_bytes(lv.G);
    if (lv.blk_err && *lv.blk_err)
        return fail(MGRIT_HIP_EHIP, "time-parallel forward solve in one launch: a device-wide barrier gave up (workgroups not resident together)");
    if (phases == 7 && lv.blk_qt && bk.first_real && !bk.project_last) {
        const dim3 grid(bk.B), block(LANES);
        if (F == 0) hipLaunchKernelGGL((blk_one_kernel<0>), grid, block, lds, e->stream, lv.dev, bk, reinterpret_cast<const double2 *>(lv.blk_qt), lv.blk_sync, lv.blk_err);
        else if (F == 2) hipLaunchKernelGGL((blk_one_kernel<2>), grid, block, lds, e->stream, lv.dev, bk, reinterpret_cast<const double2 *>(lv.blk_qt), lv.blk_sync, lv.blk_err);
        else if (F == 3) hipLaunchKernelGGL((blk_one_kernel<3>), grid, block, lds, e->stream, lv.dev, bk, reinterpret_cast<const double2 *>(lv.blk_qt), lv.blk_sync, lv.blk_err);
        else hipLaunchKernelGGL((blk_one_kernel<4>), grid, block, lds, e->stream, lv.dev, bk, reinterpret_cast<const double2 *>(lv.blk_qt), lv.blk_sync, lv.blk_err);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const int cap = 256 * blk_wgs_per_cu(lv);
    const dim3 block(lv.dev.T);
    const bool adv = bk.fourier != 0, dft = bk.fourier == 2;
    const int n = lv.dev.n, fft_threads = std::min(n / 2, 1024);
    const size_t fft_lds = (size_t)n * sizeof(double2);
    const size_t dft_lds = (size_t)n * sizeof(double2);                          // the table of all n roots of unity
    const unsigned dft_gy = (unsigned)(((n + 15) / 16 + DFT_WAVES - 1) / DFT_WAVES);   // DFT_WAVES tiles of 16 per workgroup
    if (phases & 1) {
        const dim3 grid(std::min(bk.B, cap));
        if (adv) hipLaunchKernelGGL((blk_local_kernel<MGRIT_HIP_STEPPER_ADVECTION1D, 0>), grid, block, lds, e->stream, lv.dev, bk);
        else if (F == 0) hipLaunchKernelGGL((blk_local_kernel<MGRIT_HIP_STEPPER_HEAT1D, 0>), grid, block, lds, e->stream, lv.dev, bk);
        else if (F == 2) hipLaunchKernelGGL((blk_local_kernel<MGRIT_HIP_STEPPER_HEAT1D, 2>), grid, block, lds, e->stream, lv.dev, bk);
        else if (F == 3) hipLaunchKernelGGL((blk_local_kernel<MGRIT_HIP_STEPPER_HEAT1D, 3>), grid, block, lds, e->stream, lv.dev, bk);
        else hipLaunchKernelGGL((blk_local_kernel<MGRIT_HIP_STEPPER_HEAT1D, 4>), grid, block, lds, e->stream, lv.dev, bk);
        const int cnt = bk.B - 1 + (bk.project_last ? 1 : 0);     // blocks whose amplitudes the recurrence reads
        if (adv) {   // what_b = FFT(W_b)
            if (cnt > 0 && dft) hipLaunchKernelGGL(adv_dft_fwd_kernel, dim3((cnt + 15) / 16, dft_gy), dim3(64 * DFT_WAVES), dft_lds, e->stream, lv.dev, bk, 0, cnt);
            else if (cnt > 0) hipLaunchKernelGGL(adv_fft_rows_kernel, dim3(cnt), dim3(fft_threads), fft_lds, e->stream, lv.dev, bk, 0, 0);
        } else if (cnt > 0) {   // what_b(k) = <q_k, W_b> on the matrix cores, chunk by chunk, then the chunks in order
            hipLaunchKernelGGL(blk_project_kernel, dim3((cnt + 15) / 16, (bk.r + 15) / 16, lv.G), dim3(64), 0, e->stream, bk, lv.dev.ld, lv.blk_part);
            hipLaunchKernelGGL(blk_sum_chunks_kernel, dim3(cnt), dim3(BLK_RMAX), 0, e->stream, bk, lv.G, lv.blk_part);
        }
    }
    if (phases & 2) {
        if (adv) hipLaunchKernelGGL(adv_scan_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, bk, n);
        else hipLaunchKernelGGL(blk_scan_kernel, dim3(1), dim3(BLK_RMAX), 0, e->stream, bk);
        if (bk.project_last) {   // the last point, which the next rank waits for
            if (adv && dft) hipLaunchKernelGGL(adv_dft_inv_kernel, dim3(1, dft_gy), dim3(64 * DFT_WAVES), dft_lds, e->stream, lv.dev, bk, bk.B - 1, 1);
            else if (adv) hipLaunchKernelGGL(adv_fft_rows_kernel, dim3(1), dim3(fft_threads), fft_lds, e->stream, lv.dev, bk, bk.B - 1, 1);
            else hipLaunchKernelGGL(blk_last_kernel, dim3(8, lv.G), dim3(LANES), 0, e->stream, lv.dev, bk);
        }
    }
    if (phases & 4) {
        BlkDev b2 = bk;
        b2.skip_last_row = bk.project_last;
        if (adv) {   // u[e_b] += W_b + Re(IFFT(c_b)) / n for the block ends not yet corrected, then the second pass
            const int cnt = bk.B - (bk.project_last ? 1 : 0);
            if (cnt > 0 && dft) hipLaunchKernelGGL(adv_dft_inv_kernel, dim3((cnt + 15) / 16, dft_gy), dim3(64 * DFT_WAVES), dft_lds, e->stream, lv.dev, bk, 0, cnt);
            else if (cnt > 0) hipLaunchKernelGGL(adv_fft_rows_kernel, dim3(cnt), dim3(fft_threads), fft_lds, e->stream, lv.dev, bk, 0, 1);
        }
        else hipLaunchKernelGGL(blk_correct_kernel, dim3((bk.B + 15) / 16, lv.dev.ld / 64), dim3(64), 0, e->stream, lv.dev, bk,
                                bk.B - (bk.project_last ? 1 : 0));
        const int items = bk.B;
        const dim3 grid(std::min(items, cap));
        if (adv) hipLaunchKernelGGL((blk_finish_kernel<MGRIT_HIP_STEPPER_ADVECTION1D, 0>), grid, block, lds, e->stream, lv.dev, b2);
        else if (F == 0) hipLaunchKernelGGL((blk_finish_kernel<MGRIT_HIP_STEPPER_HEAT1D, 0>), grid, block, lds, e->stream, lv.dev, b2);
        else if (F == 2) hipLaunchKernelGGL((blk_finish_kernel<MGRIT_HIP_STEPPER_HEAT1D, 2>), grid, block, lds, e->stream, lv.dev, b2);
        else if (F == 3) hipLaunchKernelGGL((blk_finish_kernel<MGRIT_HIP_STEPPER_HEAT1D, 3>), grid, block, lds, e->stream, lv.dev, b2);
        else hipLaunchKernelGGL((blk_finish_kernel<MGRIT_HIP_STEPPER_HEAT1D, 4>), grid, block, lds, e->stream, lv.dev, b2);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// process-wide cache of sine-mode tables (blk_config): the four most recent (n, ld, rows); a table in use by a level stays alive
// through the level's shared_ptr after it has left the cache
struct BlkModeKey { int dev, n, ld, rows; std::shared_ptr<double> tab; };
std::vector<BlkModeKey> &blk_mode_cache() { static std::vector<BlkModeKey> c; return c; }
std::mutex &blk_mode_lock() { static std::mutex m; return m; }
int blk_mode_device() { int d = -1; (void)hipGetDevice(&d); return d; }    // (a table lives in ONE device's memory)
std::shared_ptr<double> blk_mode_table_cached(int n, int ld, int rows) {
    const int dev = blk_mode_device();
    std::lock_guard<std::mutex> g(blk_mode_lock());
    for (auto &k : blk_mode_cache())
        if (k.dev == dev && k.n == n && k.ld == ld && k.rows >= rows) return k.tab;
    return nullptr;
}
void blk_mode_table_store(int n, int ld, int rows, const std::shared_ptr<double> &tab) {
    std::lock_guard<std::mutex> g(blk_mode_lock());
    auto &c = blk_mode_cache();
    if (c.size() >= 4) c.erase(c.begin());
    c.push_back({blk_mode_device(), n, ld, rows, tab});
}

int blk_config(mgrit_hip_engine *e, int lvl, int r, int first_real, int has_successor, double *uh_in, double *uh_out) {
    Level &lv = e->L[lvl];
    lv.blk = BlkDev{};
    lv.blk_state = 0;
    if (r == 0) return 0;
    if (lv.h2d) {   // Heat2D (backward Euler, one rank): the full spectrum, no tables here (h2d_block_build on first use)
        if (!first_real || has_successor) return fail(MGRIT_HIP_EUNSUPPORTED, "time-parallel forward solve of a Heat2D level: one rank");
        if (h2d_block_ok(lv, lvl)) lv.blk_state = 1;
        else if (r > 0) return fail(MGRIT_HIP_EUNSUPPORTED, "time-parallel forward solve of a Heat2D level: level > 0, backward Euler, 64 .. %d steps", H2D_MAX_BATCH * BLK_K);
        return 0;
    }
    const bool heat = lv.dev.kind == MGRIT_HIP_STEPPER_HEAT1D, adv = lv.dev.kind == MGRIT_HIP_STEPPER_ADVECTION1D;
    const bool can = lvl > 0 && (heat || adv) && !lv.wide && !lv.h2d && lv.dev.n_pts >= 2;
    if (r < 0 && !can) return 0;
    if (!can) return fail(MGRIT_HIP_EUNSUPPORTED, "time-parallel forward solve: a register-resident Heat1D / Advection1D level > 0");
    const int n = lv.dev.n, ld = lv.dev.ld, nt = lv.dev.n_pts;
    std::vector<double> D;
    if (r < 0) {   // one rank: the rule on the local (= global) grid
        r = heat ? blk_rank(n, lv.fac, nt, lv.t_host.data(), nullptr) : (blk_fourier_ok(n, nt) ? n : 0);
        if (r == 0) return 0;
    }
    const int B = (nt - 1) / BLK_K;   // whole blocks of the rank's share, the last one with the remainder (a rank of a sharded level may hold fewer than 4)
    if (B < 1) return fail(MGRIT_HIP_EINVAL, "time-parallel forward solve: %d local steps are fewer than one block of %d", nt - 1, BLK_K);
    if (!first_real && !uh_in) return fail(MGRIT_HIP_EINVAL, "time-parallel forward solve: a rank with a predecessor needs uh_in");
    if (has_successor && !uh_out) return fail(MGRIT_HIP_EINVAL, "time-parallel forward solve: a rank with a successor needs uh_out");
    int rc;
    BlkDev bk{};
    if (heat) {
        if (r > BLK_RMAX || r > n) return fail(MGRIT_HIP_EINVAL, "time-parallel forward solve: r = %d modes outside [1, %d]", r, std::min(BLK_RMAX, n));
        // D_b(k) = prod over the block's steps of 1 / (1 + dt_i fac 4 sin^2(theta_k / 2)), products in step order
        std::vector<double> Dt((size_t)B * BLK_RMAX, 0.0);
        for (int b = 0; b < B; ++b) {
            const int first = BLK_K * b + 1, last = b == B - 1 ? nt - 1 : BLK_K * (b + 1);
            for (int k = 0; k < BLK_RMAX && k < n; ++k) {
                const double lam = blk_lam4(n, k);
                double d = 1.0;
                for (int i = first; i <= last; ++i) d = d * (1.0 / (1.0 + ((lv.t_host[i] - lv.t_host[i - 1]) * lv.fac) * lam));
                Dt[(size_t)b * BLK_RMAX + k] = d;
            }
        }
        // the sine modes in row storage order: a table of (r rounded up to 16) x ld doubles that depends on (n, ld, rows) only -- 8 MB and
        // 0.8 M calls of sin at config 3. A process that builds solver after solver on the same spatial grid (a service; the second
        // constructor of bench.py's time-to-solution) finds it in a small process-wide cache, device resident and immutable
        const int q_rows = (r + 15) / 16 * 16;
        std::shared_ptr<double> qtab = blk_mode_table_cached(n, ld, q_rows);
        if (!qtab) {
            std::vector<double> Q((size_t)q_rows * ld, 0.0);   // (zero rows up to a multiple of 16 modes)
            const double sc = std::sqrt(2.0 / (double)(n + 1));
            {   // the mode rows are independent: on a few host threads (r n calls of sin -- 0.8 M at config 3 -- were 15 ms of a solve's setup)
                auto rows = [&](int k0, int k1) {
                    for (int k = k0; k < k1; ++k)
                        for (int j = 0; j < n; ++j) {
                            const long m = ((long)(k + 1) * (long)(j + 1)) % (2L * (n + 1));
                            Q[(size_t)k * ld + row_pos(j)] = sc * std::sin(M_PI * (double)m / (double)(n + 1));
                        }
                };
                const int rq = std::min(q_rows, n);    // (every row of the table is a mode: a later level with more modes may reuse it)
                const unsigned hw = std::thread::hardware_concurrency();
                const int nth = (size_t)rq * n < 65536 ? 1 : std::max(1, std::min({8, (int)(hw ? hw : 1), rq}));
                std::vector<std::thread> pool;
                for (int w = 1; w < nth; ++w) pool.emplace_back(rows, (int)((long)rq * w / nth), (int)((long)rq * (w + 1) / nth));
                rows(0, rq / nth);
                for (std::thread &th : pool) th.join();
            }
            double *raw = nullptr;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&raw), sizeof(double) * Q.size()));
            qtab = std::shared_ptr<double>(raw, [](double *p) { (void)hipFree(p); });
            HIP_TRY(hipMemcpyAsync(raw, Q.data(), sizeof(double) * Q.size(), hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            blk_mode_table_store(n, ld, q_rows, qtab);
        }
        lv.blk_q = qtab;
        double *dQ = qtab.get(), *dD, *dW;
        if ((rc = dev_upload(lv, e->stream, Dt, &dD))) return rc;
        // amplitudes, propagated amplitudes and the chunks' partial sums: one slab, zeroed on the device ([2 + G][B][BLK_RMAX]: 9 MB at
        // config 3 -- modes past r are never written and must read as zero)
        const size_t per = (size_t)B * BLK_RMAX, slab = (size_t)(2 + lv.G) * per;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dW), sizeof(double) * slab));
        lv.allocs.push_back(dW);
        HIP_TRY(hipMemsetAsync(dW, 0, sizeof(double) * slab, e->stream));
        bk.Q = dQ; bk.D = dD; bk.what = dW; bk.C = dW + per;
        lv.blk_part = dW + 2 * per;
    } else {
        if (r != n || !(n >= 64 && n <= BLK_FOURIER_MAX_N))
            return fail(MGRIT_HIP_EINVAL, "time-parallel forward solve of an Advection1D level: all n modes, n in [64, %d] (n = %d, r = %d)", BLK_FOURIER_MAX_N, n, r);
        // twiddles (n a power of two: the first n/2 roots of unity; else all n: orc_dft_table) and the blocks' complex propagators
        // (the oracle's orc_fft_twiddles / orc_adv_block_propagators: same expressions)
        const bool pow2 = (n & (n - 1)) == 0;
        const int n_tw = pow2 ? n / 2 : n;
        std::vector<double> W((size_t)2 * n_tw, 0.0), Dt((size_t)B * n * 2, 0.0);
        for (int t = 0; t < n_tw; ++t) {
            const double ang = 2.0 * M_PI * (double)t / (double)n;
            W[2 * (size_t)t] = sep_cos(ang); W[2 * (size_t)t + 1] = -sep_sin(ang);
        }
        for (int b = 0; b < B; ++b) {
            const int first = BLK_K * b + 1, last = b == B - 1 ? nt - 1 : BLK_K * (b + 1);
            for (int k = 0; k < n; ++k) {
                const double th = 2.0 * M_PI * (double)k / (double)n, cs = sep_cos(th), sn = sep_sin(th);
                double pr = 1.0, pi = 0.0;
                for (int i = first; i <= last; ++i) {
                    const double alpha = (lv.t_host[i] - lv.t_host[i - 1]) * lv.fac;
                    const double mr = (1.0 + alpha) - alpha * cs, mi = alpha * sn, den = mr * mr + mi * mi;
                    const double dr = mr / den, di = -mi / den;
                    const double qr = pr * dr - pi * di, qi = pr * di + pi * dr;
                    pr = qr; pi = qi;
                }
                Dt[((size_t)b * n + k) * 2] = pr; Dt[((size_t)b * n + k) * 2 + 1] = pi;
            }
        }
        double *dT, *dD, *dW;
        if ((rc = dev_upload(lv, e->stream, W, &dT))) return rc;
        if ((rc = dev_upload(lv, e->stream, Dt, &dD))) return rc;
        const size_t per = (size_t)B * n * 2;       // amplitudes and propagated amplitudes: zeroed on the device
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dW), sizeof(double) * 2 * per));
        lv.allocs.push_back(dW);
        HIP_TRY(hipMemsetAsync(dW, 0, sizeof(double) * 2 * per, e->stream));
        bk.tw = reinterpret_cast<const double2 *>(dT); bk.D = dD; bk.what = dW; bk.C = dW + per;
        bk.fourier = pow2 ? 1 : 2;                  // 1: radix-2 in LDS, 2: ordered sums on the matrix cores
        bk.lg_n = 0;
        while ((1 << bk.lg_n) < n) ++bk.lg_n;
    }
    {
        double *dWs = nullptr;   // (zeroed on the device: as a host vector of B rows it was 33 MB allocated, cleared and copied at config 3)
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dWs), sizeof(double) * (size_t)B * ld));
        lv.allocs.push_back(dWs);
        HIP_TRY(hipMemsetAsync(dWs, 0, sizeof(double) * (size_t)B * ld, e->stream));
        bk.Ws = dWs;
    }
    lv.blk_qt = nullptr;
    if (heat && lv.G == 1 && lv.dev.T == LANES && ld == GROUP && r <= BLK_ONE_MAX_R && B <= BLK_ONE_MAX_B && first_real && !has_successor &&
        blk_one_launch_enabled()) {
        // a small level: the whole solve as one launch (blk_one_kernel)
        double *qt = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&qt), sizeof(double) * (size_t)GROUP * BLK_ONE_MAX_R));
        lv.allocs.push_back(qt);
        hipLaunchKernelGGL(blk_transpose_modes_kernel, dim3(GROUP / 2), dim3(BLK_ONE_MAX_R), 0, e->stream, bk.Q, ld, (r + 15) / 16 * 16,
                           reinterpret_cast<double2 *>(qt));
        HIP_TRY(hipGetLastError());
        if (!lv.blk_sync) {
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&lv.blk_sync), 256));
            lv.allocs.push_back(lv.blk_sync);
            HIP_TRY(hipMemsetAsync(lv.blk_sync, 0, 256, e->stream));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&lv.blk_err), 256, hipHostMallocMapped));
            *lv.blk_err = 0u;
        }
        lv.blk_qt = qt;
    }
    bk.uh_in = uh_in; bk.uh_out = has_successor ? uh_out : nullptr;
    bk.r = r; bk.B = B; bk.n_steps = nt - 1;
    bk.first_real = first_real ? 1 : 0;
    bk.project_last = has_successor ? 1 : 0;
    bk.skip_last_row = 0;
    lv.blk = bk;
    return 0;
}

}  // namespace

// ===============================================================================================================
// C ABI
// ===============================================================================================================
extern "C" {

int mgrit_hip_abi_version(void) { return MGRIT_HIP_ABI_VERSION; }
const char *mgrit_hip_last_error(void) { return g_err.c_str(); }

int mgrit_hip_row_stride(int n) { return n < 1 ? 0 : ((n + GROUP - 1) / GROUP) * GROUP; }

int mgrit_hip_row_position(int n, int j) {
    if (n < 1 || j < 0 || j >= n) return -1;
    return row_pos(j);
}

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
    for (auto &lv : e->L) lv.arena = &e->arena;
    *out = e;
    return 0;
}

int mgrit_hip_destroy(mgrit_hip_engine *e) {
    if (!e) return 0;
    (void)hipStreamSynchronize(e->stream);
    for (auto &lv : e->L) {
        if (lv.blk_err) (void)hipHostFree(lv.blk_err);
        if (lv.h2d) {
            if (lv.h2d->W0) (void)hipFree(lv.h2d->W0);
            if (lv.h2d->W1) (void)hipFree(lv.h2d->W1);
            if (lv.h2d->rowsq) (void)hipFree(lv.h2d->rowsq);
            if (lv.h2d->Wc0) (void)hipFree(lv.h2d->Wc0);
            if (lv.h2d->Wc1) (void)hipFree(lv.h2d->Wc1);
            if (lv.h2d->blk.rim_flag) (void)hipHostFree(lv.h2d->blk.rim_flag);
            if (lv.h2d->blk.rim_ev) (void)hipEventDestroy(lv.h2d->blk.rim_ev);
            delete lv.h2d;
        }
        if (lv.wide) {
            for (double *p : {lv.wide->W, lv.wide->tot, lv.wide->car, lv.wide->z0, lv.wide->red, lv.wide->T0})
                if (p) (void)hipFree(p);
            delete lv.wide;
        }
        for (void *p : lv.allocs) (void)hipFree(p);
        if (lv.scratch) (void)hipFree(lv.scratch);
    }
    e->arena.release();
    if (e->chain_gran) (void)hipFree(e->chain_gran);
    if (e->sched) (void)hipFree(e->sched);
    if (e->chain_err) (void)hipHostFree(e->chain_err);
    if (e->pinned) (void)hipHostFree(e->pinned);
    for (double *p : e->pinned_old) (void)hipHostFree(p);
    if (e->ev_read) (void)hipEventDestroy(e->ev_read);
    links_close(e, false);
    if (e->xscratch) (void)hipFree(e->xscratch);
    if (e->mirror_cur) (void)hipFree(e->mirror_cur);
    for (auto &r : e->trecs) { (void)hipEventDestroy(r.ev0); (void)hipEventDestroy(r.ev1); }
    for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
    delete e;
    return 0;
}

static int chain_status(mgrit_hip_engine *e) {
    if (e->chain_err && *e->chain_err != 0u) {
        *e->chain_err = 0u;
        if (e->sched) (void)hipMemset(e->sched, 0, 1024);      // the workers left their counters behind
        if (e->chain_gran) (void)hipMemset(e->chain_gran, 0, sizeof(u64) * 4 * MAX_G * 4);
        return fail(MGRIT_HIP_EHIP, "cross-workgroup chain kernel timed out waiting for a peer workgroup (results invalid)");
    }
    return 0;
}

int mgrit_hip_sync(mgrit_hip_engine *e) {
    if (!e) return fail(MGRIT_HIP_EINVAL, "null engine");
    HIP_TRY(hipStreamSynchronize(e->stream));
    return chain_status(e);
}

int mgrit_hip_level_heat1d(mgrit_hip_engine *e, int lvl, int n_pts_local, const double *t_local, int n, int ld,
                           double fac, int K, const double *s, const double *tau) {
    return level_common(e, lvl, MGRIT_HIP_STEPPER_HEAT1D, n_pts_local, t_local, n, ld, fac, K, s, tau);
}

int mgrit_hip_level_heat1d_2pts(mgrit_hip_engine *e, int lvl, int n_pts_local, const double *t_local, int n, int ld,
                                double fac, double dtau, int order, int K, const double *s, const double *tau,
                                const double *tau2) {
    return level_heat1d_2pts(e, lvl, n_pts_local, t_local, n, ld, fac, dtau, order, K, s, tau, tau2);
}

int mgrit_hip_level_advection1d(mgrit_hip_engine *e, int lvl, int n_pts_local, const double *t_local, int n, int ld,
                                double fac) {
    return level_common(e, lvl, MGRIT_HIP_STEPPER_ADVECTION1D, n_pts_local, t_local, n, ld, fac, 0, nullptr, nullptr);
}

int mgrit_hip_level_heat2d(mgrit_hip_engine *e, int lvl, int n_pts_local, const double *t_local, int nx, int ny, int ld,
                           double fx, double fy, double theta, const double *bc, int K, const double *S, const double *tau) {
    int rc = check_level(e, lvl, false);
    if (rc) return rc;
    if (nx < 3 || ny < 3 || nx > 2050 || ny > 2050) return fail(MGRIT_HIP_EUNSUPPORTED, "Heat2D grid %dx%d outside [3,2050]^2", nx, ny);
    if (ld < nx * ny || (ld % 16) != 0) return fail(MGRIT_HIP_EINVAL, "ld=%d must be a multiple of 16 and >= nx*ny=%d", ld, nx * ny);
    if (!(theta == 0.0 || theta == 0.5 || theta == 1.0)) return fail(MGRIT_HIP_EINVAL, "theta must be 0 (FE), 0.5 (CN) or 1 (BE)");
    if (n_pts_local < 0 || (n_pts_local > 0 && !t_local) || !bc) return fail(MGRIT_HIP_EINVAL, "bad arguments");
    if (K < 0 || K > 8 || (K > 0 && (!S || (n_pts_local > 0 && !tau)))) return fail(MGRIT_HIP_EINVAL, "bad forcing description (K=%d)", K);
    Level &lv = e->L[lvl];
    if (lv.set) return fail(MGRIT_HIP_EINVAL, "level %d already described", lvl);
    H2DHost *h = new H2DHost();
    lv.h2d = h;
    H2DDev &H = h->dev;
    H.nx = nx; H.ny = ny; H.mi = nx - 2; H.mj = ny - 2;
    h->HPx = (((H.mi + 1) / 2 + 63) / 64) * 64; h->HPy = (((H.mj + 1) / 2 + 63) / 64) * 64;
    H.Mi = 2 * h->HPx; H.Mj = 2 * h->HPy;   // one padded length per axis for the natural and the spectral layout
    H.K = K; H.n_pts = n_pts_local; H.ld = ld; H.fx = fx; H.fy = fy; H.theta = theta;
    h->dts.assign(n_pts_local > 0 ? n_pts_local : 0, 0.0);
    for (int i = 1; i < n_pts_local; ++i) h->dts[i] = t_local[i] - t_local[i - 1];
    // tables
    std::vector<double> fxe, fxo, fxet, fxot, fye, fyo, fyet, fyot;
    h2d_axis_tables(H.mi, h->HPx, fx, fxe, fxo, fxet, fxot, h->lx);
    h2d_axis_tables(H.mj, h->HPy, fy, fye, fyo, fyet, fyot, h->ly);
    std::vector<double> W((size_t)H.Mi * H.Mj, 0.0), Sp((size_t)(K > 0 ? K : 0) * H.Mi * H.Mj, 0.0), bcv(bc, bc + (size_t)nx * ny);
    H.has_w = 0;
    for (int a = 0; a < H.mi; ++a)
        for (int b = 0; b < H.mj; ++b) {
            double w = 0.0;
            if (a == 0) w += fx * bc[(size_t)0 * ny + b + 1];
            if (a == H.mi - 1) w += fx * bc[(size_t)(nx - 1) * ny + b + 1];
            if (b == 0) w += fy * bc[(size_t)(a + 1) * ny + 0];
            if (b == H.mj - 1) w += fy * bc[(size_t)(a + 1) * ny + ny - 1];
            W[(size_t)a * H.Mj + b] = w;
            if (w != 0.0) H.has_w = 1;
        }
    for (int k = 0; k < K; ++k)
        for (int a = 0; a < H.mi; ++a)
            for (int b = 0; b < H.mj; ++b) Sp[((size_t)k * H.Mi + a) * H.Mj + b] = S[((size_t)k * H.mi + a) * H.mj + b];
    std::vector<double> tauv;
    if (K > 0) tauv.assign(tau, tau + (size_t)K * n_pts_local);
    double *d_bc, *d_W, *d_S, *d_tau, *d_dt;
    if ((rc = dev_upload(lv, e->stream, fxe, &h->Fxe)) || (rc = dev_upload(lv, e->stream, fxo, &h->Fxo)) ||
        (rc = dev_upload(lv, e->stream, fxet, &h->FxeT)) || (rc = dev_upload(lv, e->stream, fxot, &h->FxoT)) ||
        (rc = dev_upload(lv, e->stream, fye, &h->Fye)) || (rc = dev_upload(lv, e->stream, fyo, &h->Fyo)) ||
        (rc = dev_upload(lv, e->stream, fyet, &h->FyeT)) || (rc = dev_upload(lv, e->stream, fyot, &h->FyoT)))
        return rc;
    if ((rc = dev_upload(lv, e->stream, bcv, &d_bc))) return rc;
    if ((rc = dev_upload(lv, e->stream, W, &d_W))) return rc;
    if ((rc = dev_upload(lv, e->stream, Sp, &d_S))) return rc;
    if ((rc = dev_upload(lv, e->stream, tauv, &d_tau))) return rc;
    if ((rc = dev_upload(lv, e->stream, h->dts, &d_dt))) return rc;
    H.bc = d_bc; H.W = d_W; H.S = d_S; H.tstop = d_tau; H.dt = d_dt;
    lv.dev.kind = MGRIT_HIP_STEPPER_HEAT2D;
    lv.dev.n = nx * ny; lv.dev.ld = ld; lv.dev.T = 0; lv.dev.n_pts = n_pts_local; lv.dev.K = K; lv.dev.stream_rows = 0;
    lv.G = 0;
    lv.set = true;
    return 0;
}

int mgrit_hip_level_bind(mgrit_hip_engine *e, int lvl, double *u, double *v, double *g) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (!u && e->L[lvl].dev.n_pts > 0) return fail(MGRIT_HIP_EINVAL, "u slab is null");
    e->L[lvl].dev.u = u; e->L[lvl].dev.v = v; e->L[lvl].dev.g = g;
    return 0;
}

int mgrit_hip_level_forcing_rows(mgrit_hip_engine *e, int lvl, const double *rows) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    Level &lv = e->L[lvl];
    if (lv.dev.kind != MGRIT_HIP_STEPPER_HEAT1D) return fail(MGRIT_HIP_EUNSUPPORTED, "forcing rows: Heat1D levels only");
    if (rows && lv.dev.K != 0) return fail(MGRIT_HIP_EINVAL, "level %d already has %d separable forcing terms", lvl, lv.dev.K);
    lv.dev.fb = rows;
    return 0;
}

int mgrit_hip_heat2d_padded(mgrit_hip_engine *e, int lvl, int *Mi, int *Mj) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (!e->L[lvl].h2d || !Mi || !Mj) return fail(MGRIT_HIP_EINVAL, "level %d is not a Heat2D level", lvl);
    *Mi = e->L[lvl].h2d->dev.Mi; *Mj = e->L[lvl].h2d->dev.Mj;
    return 0;
}

int mgrit_hip_level_heat2d_forcing_rows(mgrit_hip_engine *e, int lvl, const double *rows) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    Level &lv = e->L[lvl];
    if (!lv.h2d) return fail(MGRIT_HIP_EUNSUPPORTED, "level %d is not a Heat2D level", lvl);
    if (rows && lv.h2d->dev.K != 0) return fail(MGRIT_HIP_EINVAL, "level %d already has %d separable forcing terms", lvl, lv.h2d->dev.K);
    lv.h2d->dev.fb = rows;
    return 0;
}

int mgrit_hip_chain_enable(mgrit_hip_engine *e, int lvl, int on) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    e->L[lvl].chain_overlapped = on != 0;
    return 0;
}

int mgrit_hip_chain_state_len(mgrit_hip_engine *e, int lvl, int *len_out) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (!len_out) return fail(MGRIT_HIP_EINVAL, "null output");
    *len_out = (lvl > 0 && e->L[lvl].dev.chT && e->L[lvl].chain_overlapped && !plain_chain() && !e->L[lvl].dev.fb) ? e->L[lvl].dev.ld + CHAIN_STATE_TAIL : 0;
    return 0;
}

int mgrit_hip_chain_bind(mgrit_hip_engine *e, int lvl, double *state) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    e->L[lvl].chain_state = state;
    return 0;
}

int mgrit_hip_chain_resume(mgrit_hip_engine *e, int lvl, int on) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (on && !e->L[lvl].chain_state) return fail(MGRIT_HIP_EINVAL, "level %d has no chain state bound", lvl);
    e->L[lvl].chain_resume = on != 0;
    return 0;
}

int mgrit_hip_block_solve_rank(int stepper, int n, double fac, int nt, const double *t, int *r_out) {
    if (!r_out || (nt > 0 && !t)) return fail(MGRIT_HIP_EINVAL, "null argument");
    *r_out = 0;
    if (stepper == MGRIT_HIP_STEPPER_HEAT1D) *r_out = (n >= 2 && n <= MGRIT_HIP_MAX_N && nt >= 2) ? blk_rank(n, fac, nt, t, nullptr) : 0;
    else if (stepper == MGRIT_HIP_STEPPER_ADVECTION1D) *r_out = blk_fourier_ok(n, nt) ? n : 0;
    return 0;
}

int mgrit_hip_block_solve_config(mgrit_hip_engine *e, int lvl, int r, int first_real, int has_successor, double *uh_in,
                                 double *uh_out) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(e->stream, &cs);
    if (cs != hipStreamCaptureStatusNone) return fail(MGRIT_HIP_EINVAL, "mgrit_hip_block_solve_config inside a stream capture");
    return blk_config(e, lvl, r, first_real, has_successor, uh_in, uh_out);
}

int mgrit_hip_block_solve_state(mgrit_hip_engine *e, int lvl, int *r_out) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (!r_out) return fail(MGRIT_HIP_EINVAL, "null output");
    const Level &lv = e->L[lvl];
    *r_out = lv.blk_state < 0 ? 0 : (lv.h2d ? (lv.blk_state > 0 ? lv.h2d->dev.mi * lv.h2d->dev.mj : 0) : lv.blk.r);
    return 0;
}

int mgrit_hip_block_solve_form(mgrit_hip_engine *e, int lvl, int *form_out) {
    int rc = check_level(e, lvl), r = 0;
    if (rc) return rc;
    if (!form_out) return fail(MGRIT_HIP_EINVAL, "null output");
    if ((rc = mgrit_hip_block_solve_state(e, lvl, &r))) return rc;
    const Level &lv = e->L[lvl];
    *form_out = r == 0 ? MGRIT_HIP_BLOCK_FORM_STEPS
                       : (!lv.h2d && lv.blk_qt && lv.blk.first_real && !lv.blk.project_last) ? MGRIT_HIP_BLOCK_FORM_ONE_LAUNCH
                                                                                            : MGRIT_HIP_BLOCK_FORM_PHASES;
    return 0;
}

int mgrit_hip_block_solve(mgrit_hip_engine *e, int lvl, int phases) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    Level &lv = e->L[lvl];
    if (lv.blk_state < 0 || lv.blk.r == 0) return fail(MGRIT_HIP_EINVAL, "level %d has no time-parallel forward solve configured", lvl);
    if (phases < 1 || phases > 7) return fail(MGRIT_HIP_EINVAL, "bad phase mask %d", phases);
    if ((rc = check_bound(lv, true))) return rc;
    Timed timed(e, MGRIT_HIP_T_CHAIN, lvl);
    return blk_launch(e, lv, phases);
}

int mgrit_hip_level_transfer(mgrit_hip_engine *e, int lvl, int kind) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (lvl + 1 >= e->n_levels || !e->L[lvl + 1].set) return fail(MGRIT_HIP_EINVAL, "describe level %d before its transfer", lvl + 1);
    const int nf = e->L[lvl].dev.n, nc = e->L[lvl + 1].dev.n;
    if (is_2pts(e->L[lvl]) != is_2pts(e->L[lvl + 1]) || (is_2pts(e->L[lvl]) && kind != MGRIT_HIP_TRANSFER_COPY))
        return fail(MGRIT_HIP_EUNSUPPORTED, "two-point levels pair with two-point levels through the copy transfer only");
    if (kind == MGRIT_HIP_TRANSFER_COPY) {
        if (nf != nc) return fail(MGRIT_HIP_EINVAL, "copy transfer needs equal DOFs (%d vs %d)", nf, nc);
    } else if (kind == MGRIT_HIP_TRANSFER_HEAT1D) {
        if (e->L[lvl].h2d || e->L[lvl + 1].h2d) return fail(MGRIT_HIP_EUNSUPPORTED, "Heat2D levels support the copy transfer only");
        if (nf != 2 * nc + 1) return fail(MGRIT_HIP_EINVAL, "full-weighting transfer needs n_fine = 2*n_coarse+1 (%d vs %d)", nf, nc);
    } else if (kind == MGRIT_HIP_TRANSFER_PERIODIC1D) {
        if (e->L[lvl].h2d || e->L[lvl + 1].h2d) return fail(MGRIT_HIP_EUNSUPPORTED, "Heat2D levels support the copy transfer only");
        if (nf != 2 * nc) return fail(MGRIT_HIP_EINVAL, "periodic transfer needs n_fine = 2*n_coarse (%d vs %d)", nf, nc);
    } else if (kind == MGRIT_HIP_TRANSFER_CALLER) {
        if (!e->L[lvl].h2d != !e->L[lvl + 1].h2d) return fail(MGRIT_HIP_EUNSUPPORTED, "a caller's transfer joins two Heat2D levels or two 1-D levels");
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
    rl.h_start.assign(start, start + n_runs);
    rl.h_len.assign(len, len + n_runs);
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
    pl.h_fine.assign(fine_idx, fine_idx + n_pairs);
    pl.h_coarse.assign(coarse_idx, coarse_idx + n_pairs);
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
    if (mode != MGRIT_HIP_RELAX_F && mode != MGRIT_HIP_RELAX_C && mode != MGRIT_HIP_RELAX_CHAIN && mode != MGRIT_HIP_RELAX_FC)
        return fail(MGRIT_HIP_EINVAL, "bad relax mode %d", mode);
    if ((rc = check_bound(lv, lvl > 0))) return rc;
    if (mode == MGRIT_HIP_RELAX_FC && (lvl == 0 || lv.h2d || is_2pts(lv) || weight_c != 1.0))
        return fail(MGRIT_HIP_EUNSUPPORTED, "relax mode FC: 1-D one-point steppers on a level > 0, weight 1");
    if (rl->n == 0) return 0;
    Timed timed(e, mode == MGRIT_HIP_RELAX_F ? MGRIT_HIP_T_RELAX_F : mode == MGRIT_HIP_RELAX_CHAIN ? MGRIT_HIP_T_CHAIN :
                   mode == MGRIT_HIP_RELAX_FC ? MGRIT_HIP_T_RELAX_FC : MGRIT_HIP_T_RELAX_C, lvl);
    const bool whole_chain = mode == MGRIT_HIP_RELAX_CHAIN && lvl > 0 && rl->n == 1 && rl->h_start[0] == 1 && rl->h_len[0] == lv.dev.n_pts - 1;
    // the whole level: the time-parallel form where the level qualifies (not configured yet: the engine's own rule)
    if (whole_chain && lv.blk_state < 0 && (rc = mgrit_hip_block_solve_config(e, lvl, -1, 1, 0, nullptr, nullptr))) return rc;
    if (lv.h2d && whole_chain && lv.blk_state > 0) {
        bool stepped = false;
        rc = h2d_block_solve(e, lvl, &stepped);
        if (rc || !stepped) return rc;
    }
    if (lv.h2d) return h2d_relax(e, lvl, rl, mode, weight_c);
    if (lv.wide) return wide_relax(e, lvl, rl, mode, weight_c);
    if (is_2pts(lv)) {
        const bool use_g = lvl > 0, weighted = mode == MGRIT_HIP_RELAX_C && weight_c != 1.0;
        const double w = weight_c, w1 = 1.0 - weight_c;
        const dim3 grid(persistent_grid(lv, rl->n)), block(lv.dev.T);
        const int fm = force_mode(lv);
#define RELAX2_CASE(O, F, G_, W_)                                                                                  \
    if (lv.order == O && fm == F && use_g == G_ && weighted == W_)                                                  \
        hipLaunchKernelGGL((relax2_kernel<O, F, G_, W_>), grid, block, smem2_bytes(lv.G), e->stream, lv.dev, rl->d_start, \
                           rl->d_len, rl->n, w, w1);
#define RELAX2_CASES(O, F) RELAX2_CASE(O, F, false, false) RELAX2_CASE(O, F, true, false) RELAX2_CASE(O, F, false, true) \
    RELAX2_CASE(O, F, true, true)
        FOR_EACH_2PTS(RELAX2_CASES)
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (whole_chain) {
        if (lv.blk.r > 0) {
            if (!lv.blk.first_real || lv.blk.project_last)
                return fail(MGRIT_HIP_EINVAL, "level %d is one rank's part of a sharded solve: call mgrit_hip_block_solve by phases", lvl);
            return blk_launch(e, lv, 7);
        }
    }
    if (mode == MGRIT_HIP_RELAX_CHAIN) {
        // sequential chain: one two-wave workgroup (compute + streamer) per group, exchange through global granules
        if (!e->chain_gran) {
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->chain_gran), sizeof(u64) * 4 * MAX_G * 4));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->chain_err), 256, hipHostMallocMapped));
            std::memset(e->chain_err, 0, 256);
            HIP_TRY(hipMemsetAsync(e->chain_gran, 0, sizeof(u64) * 4 * MAX_G * 4, e->stream));
        }
        if ((rc = ensure_sched(e))) return rc;
        const bool use_g = lvl > 0;
        const int fm = force_mode(lv);
        for (int r = 0; r < rl->n; ++r) {
            const int st = rl->h_start[r], ln = rl->h_len[r];
            if ((unsigned)ln > 0x10000000u) return fail(MGRIT_HIP_EUNSUPPORTED, "chain of %d steps", ln);
            int *sel = e->reserve > 0 ? e->sched : nullptr;   // chain_worker uses the words 4, 5 of the block
            const dim3 grid(sel ? 256 : 8 * lv.G), block(2 * LANES);
            if (lv.dev.chT && lv.chain_overlapped && use_g && fm <= 1 && !plain_chain()) {   // (level 0 = a one-level hierarchy: plain)   // one coefficient set, several groups: the overlapped chain
                const int resume = (lv.chain_resume && r == 0) ? 1 : 0;
                lv.chain_resume = false;
                double *state = lv.chain_state;
                if (fm == 0 && !use_g) hipLaunchKernelGGL((chain2_kernel<0, false>), grid, block, 0, e->stream, lv.dev, st, ln, e->chain_gran, e->chain_err, state, resume, e->sched, sel);
                if (fm == 0 && use_g) hipLaunchKernelGGL((chain2_kernel<0, true>), grid, block, 0, e->stream, lv.dev, st, ln, e->chain_gran, e->chain_err, state, resume, e->sched, sel);
                if (fm == 1 && !use_g) hipLaunchKernelGGL((chain2_kernel<1, false>), grid, block, 0, e->stream, lv.dev, st, ln, e->chain_gran, e->chain_err, state, resume, e->sched, sel);
                if (fm == 1 && use_g) hipLaunchKernelGGL((chain2_kernel<1, true>), grid, block, 0, e->stream, lv.dev, st, ln, e->chain_gran, e->chain_err, state, resume, e->sched, sel);
                HIP_TRY(hipGetLastError());
#ifdef MGRIT_EXPERIMENT_COUNT_SPINS
                HIP_TRY(hipStreamSynchronize(e->stream));
                std::fprintf(stderr, "chain2 polls per step:");
                for (int g = 0; g < lv.G; ++g) std::fprintf(stderr, " %.2f", (double)e->chain_err[1 + g] / ln);
                std::fprintf(stderr, "\n");
#endif
                continue;
            }
            lv.chain_resume = false;
            if (lv.G >= 2 && lv.G <= chain_local_max_g()) {   // a few groups: all workers in one workgroup, totals through LDS
#define CHAIN_LOCAL_CASE(K, F, G_)                                                                            \
    if (lv.dev.kind == K && fm == F && use_g == G_)                                                            \
    {                                                                                                          \
        if (K == MGRIT_HIP_STEPPER_ADVECTION1D && lv.G <= CHAIN_SPLIT_MAX_G)                                   \
            hipLaunchKernelGGL((chain_local_kernel<K, F, G_, true>), dim3(1), dim3(4 * lv.G * LANES), chain_local_lds(lv.G), e->stream, lv.dev, st, ln, e->chain_err); \
        else                                                                                                   \
            hipLaunchKernelGGL((chain_local_kernel<K, F, G_, false>), dim3(1), dim3(2 * lv.G * LANES), chain_local_lds(lv.G), e->stream, lv.dev, st, ln, e->chain_err); \
    }
#define CHAIN_LOCAL_CASES(K, F) CHAIN_LOCAL_CASE(K, F, false) CHAIN_LOCAL_CASE(K, F, true)
                FOR_EACH_STEPPER(CHAIN_LOCAL_CASES)
                HIP_TRY(hipGetLastError());
                continue;
            }
#define CHAIN_CASE(K, F, G_, S_)                                                                              \
    if (lv.dev.kind == K && fm == F && use_g == G_ && (lv.G == 1) == S_)                                       \
        hipLaunchKernelGGL((chain_kernel<K, F, G_, S_>), grid, dim3(3 * LANES), 0, e->stream, lv.dev, st, ln, e->chain_gran, e->chain_err, e->sched, sel);
#define CHAIN_CASES(K, F) CHAIN_CASE(K, F, false, false) CHAIN_CASE(K, F, true, false) CHAIN_CASE(K, F, false, true) \
    CHAIN_CASE(K, F, true, true)
            FOR_EACH_STEPPER(CHAIN_CASES)
            HIP_TRY(hipGetLastError());
        }
        return 0;
    }
    {
        const bool use_g = lvl > 0;
        const int role = mode == MGRIT_HIP_RELAX_F ? ROLE_F : mode == MGRIT_HIP_RELAX_FC ? ROLE_FC : (weight_c != 1.0) ? ROLE_C_WEIGHTED : ROLE_C;
        const double w = weight_c, w1 = 1.0 - weight_c;
        // persistent grid: as many workgroups as stay resident (LDS- and thread-limited), at most one per run
        const size_t lds = smem_bytes(lv.G, lv.dev.kind);
        const dim3 grid(persistent_grid(lv, rl->n)), block(lv.dev.T);
        const int fm = force_mode(lv);
#define RELAX_CASE(K, F, G_, R)                                                                                    \
    if (lv.dev.kind == K && fm == F && use_g == G_ && role == R)                                                    \
        hipLaunchKernelGGL((relax_kernel<K, F, G_, R>), grid, block, lds, e->stream, sched_dev(e, lv), rl->d_start, rl->d_len, rl->n, \
                           w, w1);
#define RELAX_CASES(K, F)                                                                                          \
    RELAX_CASE(K, F, false, ROLE_F) RELAX_CASE(K, F, true, ROLE_F) RELAX_CASE(K, F, false, ROLE_C)                  \
    RELAX_CASE(K, F, true, ROLE_C) RELAX_CASE(K, F, false, ROLE_C_WEIGHTED) RELAX_CASE(K, F, true, ROLE_C_WEIGHTED) \
    RELAX_CASE(K, F, true, ROLE_FC)
        // (one-group Heat1D levels: the F+C pass of a cycle compiled for ONE wave per workgroup, see small_wg_instances)
        const int tb = sweep_tb(lv.dev.T);
        if (tb != 1024 && lv.dev.kind == MGRIT_HIP_STEPPER_HEAT1D && use_g && role == ROLE_FC && fm <= 2) {
#define RELAX_SMALL(F, TB_)                                                                                                           \
    if (fm == F && tb == TB_) hipLaunchKernelGGL((relax_kernel<MGRIT_HIP_STEPPER_HEAT1D, F, true, ROLE_FC, TB_>), grid, block, lds, e->stream, \
                                                 sched_dev(e, lv), rl->d_start, rl->d_len, rl->n, w, w1);
            RELAX_SMALL(0, LANES) RELAX_SMALL(1, LANES) RELAX_SMALL(2, LANES) RELAX_SMALL(0, 512) RELAX_SMALL(1, 512) RELAX_SMALL(2, 512)
        } else {
            FOR_EACH_STEPPER(RELAX_CASES)
        }
        HIP_TRY(hipGetLastError());
    }
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
    Timed timed(e, MGRIT_HIP_T_RESIDUAL, lvl);
    if (lv.h2d) return h2d_points_sumsq(e, lvl, rl, nullptr, sumsq_out);
    if (lv.wide) return wide_points_sumsq(e, lvl, rl, nullptr, sumsq_out);
    if (is_2pts(lv)) LAUNCH2_BY_ORDER(residual2_kernel, lv, persistent_grid(lv, rl->n), lv.dev, rl->d_start, rl->n, sumsq_out);
    else LAUNCH_BY_KIND(residual_kernel, lv, persistent_grid(lv, rl->n), sched_dev(e, lv), rl->d_start, rl->n, sumsq_out);
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
    Timed timed(e, MGRIT_HIP_T_JUMP, lvl);
    if (lv.h2d) return h2d_points_sumsq(e, lvl, rl, prev, sumsq_out);
    if (lv.wide) return wide_points_sumsq(e, lvl, rl, prev, sumsq_out);
    if (is_2pts(lv))
        hipLaunchKernelGGL(jump2_kernel, dim3(rl->n), dim3(lv.dev.T), smem2_bytes(lv.G), e->stream, lv.dev, rl->d_start, prev, sumsq_out);
    else
        hipLaunchKernelGGL(jump_kernel, dim3(rl->n), dim3(lv.dev.T), smem_bytes(lv.G, lv.dev.kind), e->stream, lv.dev, rl->d_start, prev, sumsq_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int caller_transfer(const Level &lf, int lvl) {
    if (lf.transfer == MGRIT_HIP_TRANSFER_CALLER)
        return fail(MGRIT_HIP_EUNSUPPORTED, "the transfer between levels %d and %d is applied by the caller", lvl, lvl + 1);
    return 0;
}

int mgrit_hip_restrict_u(mgrit_hip_engine *e, int lvl, int pairs_id) {
    PairList *pl;
    int rc = get_pairs(e, lvl, pairs_id, &pl);
    if (rc) return rc;
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    if ((rc = caller_transfer(lf, lvl))) return rc;
    if ((rc = check_bound(lf, false)) || (rc = check_bound(lc, false))) return rc;
    if (pl->n == 0) return 0;
    Timed timed(e, MGRIT_HIP_T_RESTRICT, lvl);
    dim3 grid(pl->n, (lc.dev.ld + 255) / 256);
    hipLaunchKernelGGL(restrict_rows_kernel, grid, dim3(256), 0, e->stream, lf.dev.u, lf.dev.ld, lf.dev.T, pl->d_fine, lc.dev.u,
                       lc.dev.ld, lc.dev.T, pl->d_coarse, lc.dev.n, lf.transfer);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mgrit_hip_copy_u_to_v(mgrit_hip_engine *e, int lvl_coarse) {
    int rc = check_level(e, lvl_coarse);
    if (rc) return rc;
    Level &lc = e->L[lvl_coarse];
    if ((rc = check_bound(lc, true))) return rc;
    if (lc.dev.n_pts == 0) return 0;
    Timed timed(e, MGRIT_HIP_T_COPY, lvl_coarse);
    HIP_TRY(hipMemcpyAsync(lc.dev.v, lc.dev.u, sizeof(double) * (size_t)lc.dev.n_pts * lc.dev.ld, hipMemcpyDeviceToDevice, e->stream));
    return 0;
}

int mgrit_hip_fas_fine_rows(mgrit_hip_engine *e, int lvl, int pairs_id, double *rows, int ld_rows) {
    PairList *pl;
    int rc = get_pairs(e, lvl, pairs_id, &pl);
    if (rc) return rc;
    Level &lf = e->L[lvl];
    if ((rc = check_bound(lf, lvl > 0))) return rc;
    if (is_2pts(lf)) return fail(MGRIT_HIP_EUNSUPPORTED, "split FAS right-hand side: one-point steppers only");
    if (pl->n > 0 && (!rows || ld_rows < lf.dev.ld)) return fail(MGRIT_HIP_EINVAL, "rows buffer missing or narrower than the level's rows");
    if (lf.h2d) {   // Heat2D: the batched Phi, the fine half written to the caller's rows (one per pair)
        if (pl->n == 0) return 0;
        Timed timed(e, MGRIT_HIP_T_FAS_RHS, lvl);
        if (pl->h2d_rows.empty()) {
            std::vector<H2DItem> fine;
            for (int p = 0; p < pl->n; ++p) fine.push_back({pl->h_fine[p] - 1, pl->h_fine[p], p, pl->h_fine[p], pl->h_fine[p]});
            if ((rc = h2d_make_plans(e, lf, fine, pl->h2d_rows))) return rc;
            if (pl->h2d_rows.size() > (size_t)((pl->n + H2D_MAX_BATCH - 1) / H2D_MAX_BATCH))
                return fail(MGRIT_HIP_EUNSUPPORTED, "Heat2D FAS rows for a caller's transfer: one time-step size per level");
        }
        for (const H2DPlan &q : pl->h2d_rows) {
            if ((rc = h2d_phi_op(e, lf, q, lf.dev.u, rows, ld_rows, lf.dev.g, lf.dev.u, H2D_OP_FAS_FINE, lvl > 0 ? 1 : 0, 1.0))) return rc;
        }
        return 0;
    }
    if ((rc = no_wide(e->L[lvl], &e->L[lvl + 1], "FAS right-hand side around a caller's transfer"))) return rc;
    if (pl->n == 0) return 0;
    Timed timed(e, MGRIT_HIP_T_FAS_RHS, lvl);
    LAUNCH_BY_KIND(fas_fine_kernel, lf, pl->n, lf.dev, pl->d_fine, pl->d_iota, rows, ld_rows, lvl > 0 ? 1 : 0);
    return 0;
}

int mgrit_hip_fas_coarse(mgrit_hip_engine *e, int lvl, int pairs_id) {
    PairList *pl;
    int rc = get_pairs(e, lvl, pairs_id, &pl);
    if (rc) return rc;
    Level &lc = e->L[lvl + 1];
    if ((rc = check_bound(lc, true))) return rc;
    if (is_2pts(lc)) return fail(MGRIT_HIP_EUNSUPPORTED, "split FAS right-hand side: one-point steppers only");
    if (lc.h2d) {
        if (pl->n == 0) return 0;
        Timed timed(e, MGRIT_HIP_T_FAS_RHS, lvl);
        if (pl->h2d_coarse.empty()) {
            std::vector<H2DItem> coarse;
            for (int p = 0; p < pl->n; ++p) { const int j = pl->h_coarse[p]; coarse.push_back({j - 1, j, j, j, j}); }
            if ((rc = h2d_make_plans(e, lc, coarse, pl->h2d_coarse))) return rc;
        }
        for (const H2DPlan &q : pl->h2d_coarse) {
            if ((rc = h2d_phi_op(e, lc, q, lc.dev.v, lc.dev.g, lc.dev.ld, lc.dev.g, lc.dev.v, H2D_OP_FAS_COARSE, 1, 1.0))) return rc;
        }
        return 0;
    }
    if ((rc = no_wide(e->L[lvl], &e->L[lvl + 1], "FAS right-hand side around a caller's transfer"))) return rc;
    if (pl->n == 0) return 0;
    Timed timed(e, MGRIT_HIP_T_FAS_RHS, lvl);
    LAUNCH_BY_KIND(fas_coarse_kernel, lc, pl->n, lc.dev, pl->d_coarse, 0);
    return 0;
}

int mgrit_hip_fas_rhs(mgrit_hip_engine *e, int lvl, int pairs_id) {
    PairList *pl;
    int rc = get_pairs(e, lvl, pairs_id, &pl);
    if (rc) return rc;
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    if ((rc = caller_transfer(lf, lvl))) return rc;
    if ((rc = check_bound(lf, lvl > 0)) || (rc = check_bound(lc, true))) return rc;
    if (pl->n == 0) return 0;
    Timed timed(e, MGRIT_HIP_T_FAS_RHS, lvl);
    if (lf.h2d || lc.h2d) {
        if (!lf.h2d || !lc.h2d || lf.transfer != MGRIT_HIP_TRANSFER_COPY)
            return fail(MGRIT_HIP_EUNSUPPORTED, "Heat2D levels need Heat2D on both levels and the copy transfer");
        return h2d_fas_rhs(e, lvl, pl);
    }
    if (is_2pts(lf) || is_2pts(lc)) {
        if (!is_2pts(lf) || !is_2pts(lc) || lf.transfer != MGRIT_HIP_TRANSFER_COPY)
            return fail(MGRIT_HIP_EUNSUPPORTED, "two-point levels pair with two-point levels through the copy transfer only");
        if (lf.wide) { if ((rc = wide_fas_fine(e, lvl, pl, lc.dev.g, lc.dev.ld, false))) return rc; }
        else { LAUNCH2_BY_ORDER(fas_fine2_kernel, lf, pl->n, lf.dev, pl->d_fine, pl->d_coarse, lc.dev.g, lc.dev.ld, lvl > 0 ? 1 : 0); }
        if (lc.wide) return wide_fas_coarse(e, lvl, pl);
        LAUNCH2_BY_ORDER(fas_coarse2_kernel, lc, pl->n, lc.dev, pl->d_coarse);
        return 0;
    }
    if (lf.transfer == MGRIT_HIP_TRANSFER_COPY) {
        if (lf.wide) { if ((rc = wide_fas_fine(e, lvl, pl, lc.dev.g, lc.dev.ld, false))) return rc; }
        else { LAUNCH_BY_KIND(fas_fine_kernel, lf, pl->n, lf.dev, pl->d_fine, pl->d_coarse, lc.dev.g, lc.dev.ld, lvl > 0 ? 1 : 0); }
    } else {
        if (lf.scratch_rows < (size_t)pl->n) {
            // a bigger list than any before: a NEW slab; the old one stays alive until the engine goes (lf.allocs) -- a captured
            // cycle that ran this sweep on a block's shorter list still launches with its address (found by the state fuzz: a
            // whole-level fas_residual by hand between two replays of a two-block plan freed the slab under the graph)
            if (lf.scratch) lf.allocs.push_back(lf.scratch);
            lf.scratch = nullptr;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&lf.scratch), sizeof(double) * (size_t)pl->n * lf.dev.ld));
            lf.scratch_rows = pl->n;
        }
        if (lf.wide) { if ((rc = wide_fas_fine(e, lvl, pl, lf.scratch, lf.dev.ld, true))) return rc; }
        else { LAUNCH_BY_KIND(fas_fine_kernel, lf, pl->n, lf.dev, pl->d_fine, pl->d_iota, lf.scratch, lf.dev.ld, lvl > 0 ? 1 : 0); }
        dim3 grid(pl->n, (lc.dev.ld + 255) / 256);
        hipLaunchKernelGGL(restrict_rows_kernel, grid, dim3(256), 0, e->stream, lf.scratch, lf.dev.ld, lf.dev.T, pl->d_iota,
                           lc.dev.g, lc.dev.ld, lc.dev.T, pl->d_coarse, lc.dev.n, lf.transfer);
        HIP_TRY(hipGetLastError());
    }
    if (lc.wide) return wide_fas_coarse(e, lvl, pl);
    LAUNCH_BY_KIND(fas_coarse_kernel, lc, pl->n, lc.dev, pl->d_coarse, 0);
    return 0;
}

int mgrit_hip_triples_create(mgrit_hip_engine *e, int lvl, int n, const int32_t *fine_idx, const int32_t *prev_fine_idx,
                             const int32_t *coarse_idx, int *id_out) {
    int rc = mgrit_hip_pairs_create(e, lvl, n, fine_idx, coarse_idx, id_out);
    if (rc) return rc;
    Level &lv = e->L[lvl];
    for (int p = 0; p < n; ++p)
        if (!prev_fine_idx || prev_fine_idx[p] < 0 || prev_fine_idx[p] >= fine_idx[p] || fine_idx[p] < 1)
            return fail(MGRIT_HIP_EINVAL, "triple %d: previous fine C slot must be a local slot below the fine slot", p);
    std::vector<int32_t> hp(prev_fine_idx, prev_fine_idx + n);
    return dev_upload(lv, e->stream, hp, &lv.pairs[*id_out].d_prev);
}

int mgrit_hip_fas_fused(mgrit_hip_engine *e, int lvl, int triples_id) { return mgrit_hip_fas_fused_opts(e, lvl, triples_id, 0); }

int mgrit_hip_fas_fused_opts(mgrit_hip_engine *e, int lvl, int triples_id, int opts) {
    PairList *pl;
    int rc = get_pairs(e, lvl, triples_id, &pl);
    if (rc) return rc;
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    if ((rc = check_bound(lf, lvl > 0)) || (rc = check_bound(lc, true))) return rc;
    if ((rc = no_wide(lf, &lc, "fused FAS residual"))) return rc;
    if (!pl->d_prev) return fail(MGRIT_HIP_EINVAL, "list %d was not created by mgrit_hip_triples_create", triples_id);
    if (lf.h2d || lc.h2d || is_2pts(lf) || is_2pts(lc) || lf.transfer != MGRIT_HIP_TRANSFER_COPY || lf.dev.kind != lc.dev.kind ||
        force_mode(lf) != force_mode(lc) || lf.dev.n != lc.dev.n)
        return fail(MGRIT_HIP_EUNSUPPORTED, "fused FAS residual needs the copy transfer and like steppers on both levels");
    if (opts & ~3) return fail(MGRIT_HIP_EINVAL, "unknown option bits %d", opts);
    if ((opts & MGRIT_HIP_FAS_WITH_F_RELAX) && lvl == 0) return fail(MGRIT_HIP_EUNSUPPORTED, "FAS sweep with its F-relaxation: levels > 0");
    if (pl->n == 0) return 0;
    Timed timed(e, (opts & MGRIT_HIP_FAS_WITH_F_RELAX) ? MGRIT_HIP_T_F_FAS : MGRIT_HIP_T_FAS_FUSED, lvl);
    const int use_g = lvl > 0 ? 1 : 0;
    if (lf.dev.kind == MGRIT_HIP_STEPPER_HEAT1D && force_mode(lf) != 3) {   // one pass per C-point, coarse tables from L2
        const dim3 grid(persistent_grid(lf, pl->n)), block(lf.dev.T);
        const int fm = force_mode(lf);
        // forcing factors of both levels are streamed (FORCE 2, the same fma per term): one Phi per point does not pay for
        // keeping them in registers, and the registers are needed for the partial g that stays live across the coarse Phi
        // one forcing term: its space factor lives in LDS (FORCE 4, closed-form Phi); bit 2 tells the kernel that the coarse
        // level's factor is the same vector (same spatial grid, same rhs), so the coarse Phi takes it from there too
        if (lf.same_factor_below < 0) lf.same_factor_below = (fm == 1 && lf.s_host == lc.s_host) ? 1 : 0;   // (131 KB compared once)
        const int kopts = opts | (lf.same_factor_below ? 4 : 0);
        const int tb = sweep_tb(lf.dev.T);
#define FAS1_CASE(F_, P_)                                                                                                  \
    if ((fm == 0 ? 0 : fm == 1 ? 4 : 2) == F_ && ((opts & MGRIT_HIP_FAS_WITH_F_RELAX) != 0) == P_) {                        \
        if (tb == LANES)                                                                                                   \
            hipLaunchKernelGGL((fas_fused1_kernel<F_, P_, LANES>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), \
                               lc.dev, pl->d_fine, pl->d_prev, pl->d_coarse, pl->n, use_g, kopts);                          \
        else if (tb == 512)                                                                                                \
            hipLaunchKernelGGL((fas_fused1_kernel<F_, P_, 512>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), \
                               lc.dev, pl->d_fine, pl->d_prev, pl->d_coarse, pl->n, use_g, kopts);                          \
        else                                                                                                               \
            hipLaunchKernelGGL((fas_fused1_kernel<F_, P_>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), \
                               lc.dev, pl->d_fine, pl->d_prev, pl->d_coarse, pl->n, use_g, kopts);                          \
    }
        FAS1_CASE(0, false) FAS1_CASE(2, false) FAS1_CASE(4, false) FAS1_CASE(0, true) FAS1_CASE(2, true) FAS1_CASE(4, true)
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (opts) return fail(MGRIT_HIP_EUNSUPPORTED, "fused FAS residual with options: Heat1D levels (one-pass form) only");
    const bool mid = lf.dev.T <= 512 && mid_wg_instances();
#define FUSED_CASE(K_, F_)                                                                                         \
    if (lf.dev.kind == K_ && force_mode(lf) == F_) {                                                                \
        if (mid)                                                                                                   \
            hipLaunchKernelGGL((fas_fused_kernel<K_, F_, 512>), dim3(persistent_grid(lf, pl->n)), dim3(lf.dev.T), smem_bytes(lf.G, lf.dev.kind), \
                               e->stream, lf.dev, lc.dev, pl->d_fine, pl->d_prev, pl->d_coarse, pl->n, use_g);       \
        else                                                                                                       \
            hipLaunchKernelGGL((fas_fused_kernel<K_, F_>), dim3(persistent_grid(lf, pl->n)), dim3(lf.dev.T), smem_bytes(lf.G, lf.dev.kind), \
                               e->stream, lf.dev, lc.dev, pl->d_fine, pl->d_prev, pl->d_coarse, pl->n, use_g);       \
    }
    FOR_EACH_STEPPER(FUSED_CASE)
    HIP_TRY(hipGetLastError());
    return 0;
}

int mgrit_hip_copy_pairs_u_to_v(mgrit_hip_engine *e, int lvl, int pairs_id) {
    PairList *pl;
    int rc = get_pairs(e, lvl, pairs_id, &pl);
    if (rc) return rc;
    Level &lc = e->L[lvl + 1];
    if ((rc = check_bound(lc, true))) return rc;
    if (pl->n == 0) return 0;
    Timed timed(e, MGRIT_HIP_T_COPY, lvl);
    dim3 grid(pl->n, (lc.dev.ld + 255) / 256);
    hipLaunchKernelGGL(restrict_rows_kernel, grid, dim3(256), 0, e->stream, lc.dev.u, lc.dev.ld, lc.dev.T, pl->d_coarse, lc.dev.v,
                       lc.dev.ld, lc.dev.T, pl->d_coarse, lc.dev.n, MGRIT_HIP_TRANSFER_COPY);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int interp_common(mgrit_hip_engine *e, int lvl, int pairs_id, int mode) {
    PairList *pl;
    int rc = get_pairs(e, lvl, pairs_id, &pl);
    if (rc) return rc;
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    if ((rc = caller_transfer(lf, lvl))) return rc;
    if ((rc = check_bound(lf, false)) || (rc = check_bound(lc, mode == 1))) return rc;
    if (pl->n == 0) return 0;
    Timed timed(e, mode == 1 ? MGRIT_HIP_T_ERROR_CORRECTION : MGRIT_HIP_T_INTERPOLATE, lvl);
    dim3 grid(pl->n, (lf.dev.ld + 255) / 256);
    hipLaunchKernelGGL(interp_rows_kernel, grid, dim3(256), 0, e->stream, lf.dev.u, lf.dev.ld, lf.dev.T, pl->d_fine, lc.dev.u,
                       lc.dev.v, lc.dev.ld, lc.dev.T, pl->d_coarse, lf.dev.n, lc.dev.n, lf.transfer, mode, (double *)nullptr, 0);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mgrit_hip_error_correction(mgrit_hip_engine *e, int lvl, int pairs_id) { return interp_common(e, lvl, pairs_id, 1); }
int mgrit_hip_interpolate(mgrit_hip_engine *e, int lvl, int pairs_id) { return interp_common(e, lvl, pairs_id, 0); }

// the truncated solves of a level whose Phi is a batch of launches over rows (Heat2D, wide 1-D states, wide two-point pairs; round 4):
// row p of a work slab starts as the OLD u of the point k-1 back (or the first point) and takes the steps up to p one batch per
// step distance -- every point at once, x = g_i + Phi(x) --; then the rows go back into u. The arithmetic of every point is the
// kernel at_kernel's: the same steps on the same values.
static int at_batched(mgrit_hip_engine *e, int lvl, int k) {
    Level &lv = e->L[lvl];
    int rc;
    const int n_pts = lv.dev.n_pts, ld = lv.dev.ld;
    Level::AtPlans &ap = lv.at_plans[k];
    if (ap.steps.empty() && ap.count == 0) {
        std::vector<int32_t> src, own;
        for (int p = 1; p < n_pts; ++p) { src.push_back(p - k + 1 > 0 ? p - k + 1 : 0); own.push_back(p); }
        ap.count = (int)own.size();
        if ((rc = dev_upload(lv, e->stream, src, &ap.d_src)) || (rc = dev_upload(lv, e->stream, own, &ap.d_own))) return rc;
        for (int d = 1; d < k; ++d) {
            std::vector<H2DItem> items;
            for (int p = 1; p < n_pts; ++p) {
                const int i = (p - k + 1 > 0 ? p - k + 1 : 0) + d;
                if (i <= p) items.push_back({p, i, p, i, i});
            }
            std::vector<H2DPlan> plans;
            if (!items.empty() && (rc = lv.h2d ? h2d_make_plans(e, lv, items, plans) : wide_make_plans(e, lv, items, plans))) return rc;
            ap.steps.push_back(plans);
        }
    }
    double *X = lv.scratch;
    const dim3 grid(ap.count, (ld + 255) / 256);
    hipLaunchKernelGGL(restrict_rows_kernel, grid, dim3(256), 0, e->stream, lv.dev.u, ld, lv.dev.T, ap.d_src, X, ld, lv.dev.T, ap.d_own,
                       lv.dev.n, MGRIT_HIP_TRANSFER_COPY);
    for (const std::vector<H2DPlan> &plans : ap.steps)
        for (const H2DPlan &pl : plans) {
            if (lv.h2d) {
                if ((rc = h2d_phi_op(e, lv, pl, X, X, ld, lv.dev.g, X, H2D_OP_F, 1, 1.0))) return rc;
            } else {
                if ((rc = wide_phi(e, lv, pl, X))) return rc;
                if ((rc = wide_finish(e, lv, pl, X, ld, lv.dev.g, X, WIDE_OP_F, 1, 1.0))) return rc;
            }
        }
    hipLaunchKernelGGL(restrict_rows_kernel, grid, dim3(256), 0, e->stream, X, ld, lv.dev.T, ap.d_own, lv.dev.u, ld, lv.dev.T, ap.d_own,
                       lv.dev.n, MGRIT_HIP_TRANSFER_COPY);
    HIP_TRY(hipGetLastError());
    return 0;
}

int mgrit_hip_at_solve(mgrit_hip_engine *e, int lvl, int k) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    Level &lv = e->L[lvl];
    if (k < 1) return fail(MGRIT_HIP_EINVAL, "distance k=%d must be at least 1", k);
    if (lvl == 0) return fail(MGRIT_HIP_EINVAL, "the truncated solve runs on a coarse level (it uses g)");
    if ((rc = check_bound(lv, true))) return rc;
    if (lv.dev.n_pts < 2) return 0;
    Timed timed(e, MGRIT_HIP_T_AT, lvl);
    const size_t rows = (size_t)lv.dev.n_pts;
    if (lv.scratch_rows < rows) {
        if (lv.scratch) lv.allocs.push_back(lv.scratch);   // (kept until the engine goes: see mgrit_hip_fas_rhs)
        lv.scratch = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&lv.scratch), sizeof(double) * rows * lv.dev.ld));
        lv.scratch_rows = rows;
    }
    if (lv.h2d || lv.wide) return at_batched(e, lvl, k);
    HIP_TRY(hipMemcpyAsync(lv.scratch, lv.dev.u, sizeof(double) * rows * lv.dev.ld, hipMemcpyDeviceToDevice, e->stream));
    if (is_2pts(lv)) { LAUNCH2_BY_ORDER(at2_kernel, lv, persistent_grid(lv, lv.dev.n_pts - 1), lv.dev, lv.scratch, k); }
    else { LAUNCH_BY_KIND(at_kernel, lv, persistent_grid(lv, lv.dev.n_pts - 1), lv.dev, lv.scratch, k); }
    return 0;
}

int mgrit_hip_ec_runs_create(mgrit_hip_engine *e, int lvl, int n_runs, const int32_t *start, const int32_t *len,
                             const int32_t *coarse_idx, int *id_out) {
    int rc = mgrit_hip_runs_create(e, lvl, n_runs, start, len, id_out);
    if (rc) return rc;
    if (lvl + 1 >= e->n_levels || !e->L[lvl + 1].set) return fail(MGRIT_HIP_EINVAL, "level %d has no coarser level", lvl);
    if (n_runs > 0 && !coarse_idx) return fail(MGRIT_HIP_EINVAL, "null coarse index list");
    Level &lv = e->L[lvl];
    for (int r = 0; r < n_runs; ++r)
        if (coarse_idx[r] < -1 || coarse_idx[r] >= e->L[lvl + 1].dev.n_pts)
            return fail(MGRIT_HIP_EINVAL, "run %d: coarse slot %d outside [-1,%d)", r, coarse_idx[r], e->L[lvl + 1].dev.n_pts);
    std::vector<int32_t> h(coarse_idx, coarse_idx + n_runs);
    return dev_upload(lv, e->stream, h, &lv.runs[*id_out].d_ec);
}

int mgrit_hip_ec_relax(mgrit_hip_engine *e, int lvl, int ec_runs_id) {
    RunList *rl;
    int rc = get_runs(e, lvl, ec_runs_id, &rl);
    if (rc) return rc;
    if (!rl->d_ec) return fail(MGRIT_HIP_EINVAL, "list %d was not created by mgrit_hip_ec_runs_create", ec_runs_id);
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    if ((rc = check_bound(lf, lvl > 0)) || (rc = check_bound(lc, true))) return rc;
    if (lf.h2d || lc.h2d || is_2pts(lf) || is_2pts(lc) || lf.transfer != MGRIT_HIP_TRANSFER_COPY || lf.dev.n != lc.dev.n)
        return fail(MGRIT_HIP_EUNSUPPORTED, "fused correction + F-relaxation needs 1-D steppers and the copy transfer");
    if ((rc = no_wide(lf, &lc, "fused correction + F-relaxation"))) return rc;
    if (rl->n == 0) return 0;
    Timed timed(e, MGRIT_HIP_T_EC_RELAX, lvl);
    const bool use_g = lvl > 0;
    const int fm = force_mode(lf);
    const dim3 grid(persistent_grid(lf, rl->n)), block(lf.dev.T);
#define ECF_CASE(K, F, G_)                                                                                          \
    if (lf.dev.kind == K && fm == F && use_g == G_)                                                                 \
        hipLaunchKernelGGL((ecf_kernel<K, F, G_>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), lc.dev, rl->d_start,  \
                           rl->d_len, rl->d_ec, rl->n);
#define ECF_CASES(K, F) ECF_CASE(K, F, false) ECF_CASE(K, F, true)
    const int tb = sweep_tb(lf.dev.T);
    if (tb != 1024 && lf.dev.kind == MGRIT_HIP_STEPPER_HEAT1D && use_g && fm <= 2) {
#define ECF_SMALL(F, TB_)                                                                                                             \
    if (fm == F && tb == TB_) hipLaunchKernelGGL((ecf_kernel<MGRIT_HIP_STEPPER_HEAT1D, F, true, TB_>), grid, block, smem_bytes(lf.G, lf.dev.kind), \
                                                 e->stream, sched_dev(e, lf), lc.dev, rl->d_start, rl->d_len, rl->d_ec, rl->n);
        ECF_SMALL(0, LANES) ECF_SMALL(1, LANES) ECF_SMALL(2, LANES) ECF_SMALL(0, 512) ECF_SMALL(1, 512) ECF_SMALL(2, 512)
    } else {
        FOR_EACH_STEPPER(ECF_CASES)
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// Host read-back of per-run scalars without any copy command: the kernels store straight into pinned, device-mapped
// host memory and the host spins on an event recorded behind the kernel. (A D2H copy command issued after a long mostly
// idle stretch -- the single-workgroup coarsest-level chain -- was observed to start ~46 ms late on MI355X/ROCm 7.2.)
static int ensure_pinned(mgrit_hip_engine *e, int n) {
    if (e->pinned_len < (size_t)n) {
        if (e->pinned) e->pinned_old.push_back(e->pinned);   // kept until the engine goes: a captured cycle may still write there
        e->pinned = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->pinned), sizeof(double) * (size_t)n, hipHostMallocMapped));
        e->pinned_len = n;
    }
    if (!e->ev_read) HIP_TRY(hipEventCreateWithFlags(&e->ev_read, hipEventDisableTiming));
    return 0;
}

static int wait_pinned(mgrit_hip_engine *e, int n, double *host) {
    HIP_TRY(hipEventRecord(e->ev_read, e->stream));
    for (;;) {  // spin: the convergence check sits on the critical path of every MGRIT iteration
        const hipError_t q = hipEventQuery(e->ev_read);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return fail(MGRIT_HIP_EHIP, "hipEventQuery: %s", hipGetErrorString(q));
    }
    std::memcpy(host, e->pinned, sizeof(double) * (size_t)n);
    return chain_status(e);
}

int mgrit_hip_residual_host(mgrit_hip_engine *e, int lvl, int runs_id, double *sumsq_host) {
    RunList *rl;
    int rc = get_runs(e, lvl, runs_id, &rl);
    if (rc) return rc;
    if (rl->n == 0) return 0;
    if (!sumsq_host) return fail(MGRIT_HIP_EINVAL, "null output");
    if ((rc = ensure_pinned(e, rl->n))) return rc;
    if ((rc = mgrit_hip_residual(e, lvl, runs_id, e->pinned))) return rc;
    return wait_pinned(e, rl->n, sumsq_host);
}

int mgrit_hip_jump_host(mgrit_hip_engine *e, int lvl, int runs_id, const double *prev, double *sumsq_host) {
    RunList *rl;
    int rc = get_runs(e, lvl, runs_id, &rl);
    if (rc) return rc;
    if (rl->n == 0) return 0;
    if (!sumsq_host) return fail(MGRIT_HIP_EINVAL, "null output");
    if ((rc = ensure_pinned(e, rl->n))) return rc;
    if ((rc = mgrit_hip_jump(e, lvl, runs_id, prev, e->pinned))) return rc;
    return wait_pinned(e, rl->n, sumsq_host);
}

int mgrit_hip_set_timing(mgrit_hip_engine *e, int enabled) {
    if (!e) return fail(MGRIT_HIP_EINVAL, "null engine");
    e->timing = enabled != 0;
    return 0;
}

int mgrit_hip_last_kernel_ms(mgrit_hip_engine *e, float *ms) {
    if (!e || !ms) return fail(MGRIT_HIP_EINVAL, "null argument");
    if (!e->last0) return fail(MGRIT_HIP_EINVAL, "no timed launch recorded");
    HIP_TRY(hipEventSynchronize(e->last1));
    HIP_TRY(hipEventElapsedTime(ms, e->last0, e->last1));
    return 0;
}

int mgrit_hip_timing_drain(mgrit_hip_engine *e, int max_records, int *kind, int *lvl, float *ms, int *n_out) {
    if (!e || !n_out || max_records < 0 || (max_records > 0 && (!kind || !lvl || !ms))) return fail(MGRIT_HIP_EINVAL, "bad arguments");
    int n = 0;
    for (auto &r : e->trecs) {
        if (n < max_records) {
            HIP_TRY(hipEventSynchronize(r.ev1));
            HIP_TRY(hipEventElapsedTime(&ms[n], r.ev0, r.ev1));
            kind[n] = r.kind; lvl[n] = r.lvl;
            ++n;
        }
        e->ev_pool.push_back(r.ev0); e->ev_pool.push_back(r.ev1);
    }
    e->trecs.clear();
    e->last0 = e->last1 = nullptr;
    *n_out = n;
    return 0;
}

int mgrit_hip_intervals_create(mgrit_hip_engine *e, int lvl, int n, const int32_t *cstart, const int32_t *cend,
                               const int32_t *cstart_coarse, const int32_t *cend_coarse, const int32_t *res_pos, int res_len,
                               int chunk, const int32_t *keep, int *id_out) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (lvl + 1 >= e->n_levels || !e->L[lvl + 1].set) return fail(MGRIT_HIP_EINVAL, "level %d has no described coarser level", lvl);
    if (n < 0 || chunk < MGRIT_HIP_CHUNK_LONG || res_len < n || !id_out || (n > 0 && (!cstart || !cend || !cstart_coarse || !cend_coarse || !res_pos)))
        return fail(MGRIT_HIP_EINVAL, "bad interval list");
    Level &lv = e->L[lvl];
    const Level &lc = e->L[lvl + 1];
    if (chunk <= 0) {   // by the level's size: walking several intervals in a row saves a row and a Phi per interval joined, but a
                        // level with fewer intervals than the chip holds workgroups wants every one of them running at once
        const bool longer = chunk == MGRIT_HIP_CHUNK_LONG;
        const int per_slot = res_len / (2 * 256 * wgs_per_cu(lv));
        chunk = per_slot >= 4 ? 4 : per_slot >= 2 ? 2 : 1;
        // (round 5, config 3's 16384 intervals through cfas_kernel / ecfr_kernel: chunks of 4 / 8 / 16 / 32 -> 6.67 / 6.61 / 6.46 / 6.62 ms
        // per cycle -- a chunk's start costs the way down a row and ~5 us, the way up two rows; at 32 a workgroup has two chunks and
        // the launch a tail. The general passes of config 5 were slower with 8 than with 4: 5.19 against 4.99 ms)
        if (longer) chunk = per_slot >= 32 ? 16 : per_slot >= 16 ? 8 : chunk;
        static const int forced = [] { const char *s = std::getenv("MGRIT_HIP_CHUNK"); return s ? std::atoi(s) : 0; }();   // (measurement switch)
        if (forced > 0) chunk = forced;
    }
    for (int i = 0; i < n; ++i) {
        if (cstart[i] < 0 || cend[i] >= lv.dev.n_pts || cend[i] - cstart[i] < 2)
            return fail(MGRIT_HIP_EINVAL, "interval %d = (%d,%d]: two C-points of the local grid with an F-point between them", i, cstart[i], cend[i]);
        if (cstart_coarse[i] < -1 || cstart_coarse[i] >= lc.dev.n_pts || cend_coarse[i] < 0 || cend_coarse[i] >= lc.dev.n_pts ||
            (cstart_coarse[i] >= 0 && cstart[i] < 1) || res_pos[i] < 0 || res_pos[i] >= res_len)
            return fail(MGRIT_HIP_EINVAL, "interval %d: coarse slot or residual position out of range", i);
    }
    std::vector<int32_t> cf, cl, cc;
    for (int i = 0; i < n;) {   // chunks: runs of consecutive intervals, cut where res_pos is a multiple of `chunk` -- a position
                                // of the LEVEL, so every list of the level (one block of a planned cycle, all intervals) is cut at
                                // the same C-points and at its own ends
        int len = 1;
        while (i + len < n && cstart[i + len] == cend[i + len - 1] && res_pos[i + len] % chunk != 0) ++len;
        cf.push_back(i); cl.push_back(len); cc.push_back(cstart_coarse[i]);
        i += len;
    }
    // what the closing C-point of every interval needs on the coarse level: the caller's word (null: everything), plus v where
    // a chunk ends: the next chunk -- of this list or of the list that continues it -- starts from that C-point, and
    // ecfr_kernel reads the old value of a chunk's first C-point from v (its fine row is being overwritten)
    std::vector<int32_t> kp(n, 3);
    if (keep)
        for (int i = 0; i < n; ++i) kp[i] = keep[i] & 3;
    for (size_t k = 0; k < cf.size(); ++k) kp[cf[k] + cl[k] - 1] |= 2;   // a chunk (of this or of the next list) starts from its end
    IntervalsDev d{};
    int32_t *p[8];
    const std::vector<int32_t> hs[8] = {std::vector<int32_t>(cstart, cstart + n), std::vector<int32_t>(cend, cend + n),
                                        std::vector<int32_t>(cend_coarse, cend_coarse + n), std::vector<int32_t>(res_pos, res_pos + n),
                                        cf, cl, cc, kp};
    for (int k = 0; k < 8; ++k)
        if ((rc = dev_upload(lv, e->stream, hs[k], &p[k]))) return rc;
    d.cstart = p[0]; d.cend = p[1]; d.cend_coarse = p[2]; d.res_pos = p[3]; d.chunk_first = p[4]; d.chunk_len = p[5];
    d.chunk_start_coarse = p[6]; d.keep = p[7]; d.n_chunks = (int)cf.size();
    lv.ivals.push_back(d);
    lv.ivals_n.push_back(res_len);
    lv.ivals_cnt.push_back(n);
    *id_out = (int)lv.ivals.size() - 1;
    return 0;
}

static int fused_level_check(mgrit_hip_engine *e, int lvl, int ivals_id, const char *what, bool level0_only) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (lvl != 0 && level0_only) return fail(MGRIT_HIP_EUNSUPPORTED, "%s: level 0 only", what);
    if (lvl + 1 >= e->n_levels || !e->L[lvl + 1].set) return fail(MGRIT_HIP_EINVAL, "level %d has no coarser level", lvl);
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    if ((rc = no_wide(lf, &lc, what))) return rc;
    if (ivals_id < 0 || ivals_id >= (int)lf.ivals.size()) return fail(MGRIT_HIP_EINVAL, "bad interval-list id %d on level %d", ivals_id, lvl);
    if ((rc = check_bound(lf, lvl > 0)) || (rc = check_bound(lc, true))) return rc;
    if (lf.dev.kind != MGRIT_HIP_STEPPER_HEAT1D || lc.dev.kind != MGRIT_HIP_STEPPER_HEAT1D || lf.transfer != MGRIT_HIP_TRANSFER_COPY ||
        lf.dev.n != lc.dev.n || force_mode(lf) != force_mode(lc) || force_mode(lf) == 3)
        return fail(MGRIT_HIP_EUNSUPPORTED, "%s needs Heat1D with separable forcing on both levels and the copy transfer", what);
    return 0;
}

int mgrit_hip_cf_fas(mgrit_hip_engine *e, int lvl, int ivals_id, int pre_relaxed) {
    int rc = fused_level_check(e, lvl, ivals_id, "fused C-F-relaxation + FAS residual", true);
    if (rc) return rc;
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    const IntervalsDev &I = lf.ivals[ivals_id];
    if (I.n_chunks == 0) return 0;
    Timed timed(e, MGRIT_HIP_T_CF_FAS, lvl);
    const dim3 grid(persistent_grid(lf, I.n_chunks)), block(lf.dev.T);
    const int tb = sweep_tb(lf.dev.T);   // (one wave per state / up to 512 threads: the instances compiled for it)
    if (force_mode(lf) == 0 && tb == LANES) hipLaunchKernelGGL((cfas_kernel<0, LANES>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), lc.dev, I, pre_relaxed ? 1 : 0);
    else if (tb == LANES) hipLaunchKernelGGL((cfas_kernel<2, LANES>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), lc.dev, I, pre_relaxed ? 1 : 0);
    else if (force_mode(lf) == 0 && tb == 512) hipLaunchKernelGGL((cfas_kernel<0, 512>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), lc.dev, I, pre_relaxed ? 1 : 0);
    else if (tb == 512) hipLaunchKernelGGL((cfas_kernel<2, 512>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), lc.dev, I, pre_relaxed ? 1 : 0);
    else if (force_mode(lf) == 0) hipLaunchKernelGGL((cfas_kernel<0>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), lc.dev, I, pre_relaxed ? 1 : 0);
    else hipLaunchKernelGGL((cfas_kernel<2>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), lc.dev, I, pre_relaxed ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int ec_relax_res_impl(mgrit_hip_engine *e, int lvl, int ivals_id, int store_all_f, double *out_caller);

int mgrit_hip_ec_relax_res(mgrit_hip_engine *e, int lvl, int ivals_id, int store_all_f) {
    return ec_relax_res_impl(e, lvl, ivals_id, store_all_f, nullptr);
}

int mgrit_hip_ec_relax_res_to(mgrit_hip_engine *e, int lvl, int ivals_id, int store_all_f, double *sumsq_out) {
    if (!sumsq_out) return fail(MGRIT_HIP_EINVAL, "null output");
    return ec_relax_res_impl(e, lvl, ivals_id, store_all_f, sumsq_out);
}

static int ec_relax_res_impl(mgrit_hip_engine *e, int lvl, int ivals_id, int store_all_f, double *out_caller) {
    int rc = fused_level_check(e, lvl, ivals_id, "fused correction + F-relaxation + residual", false);
    if (rc) return rc;
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    const IntervalsDev &I = lf.ivals[ivals_id];
    if (I.n_chunks == 0) return 0;
    const dim3 grid(persistent_grid(lf, I.n_chunks)), block(lf.dev.T);
    if (lvl > 0) {   // coarser level: rows of g, every F-point stored (the finer level's correction reads them), no residual
        Timed timed(e, MGRIT_HIP_T_EC_RELAX, lvl);
        const int fme = force_mode(lf) == 0 ? 0 : force_mode(lf) == 1 ? 4 : 2;   // (one term: its space factor in LDS)
        const int tb = sweep_tb(lf.dev.T);
#define ECFR_UP(F, ...)                                                                                                               \
    if (fme == F) hipLaunchKernelGGL((ecfr_kernel<F, true, false, ##__VA_ARGS__>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, \
                                     sched_dev(e, lf), lc.dev, I, (double *)nullptr, 1, (double *const *)nullptr, 0);
        if (tb == LANES) { ECFR_UP(0, LANES) ECFR_UP(4, LANES) ECFR_UP(2, LANES) }
        else if (tb == 512) { ECFR_UP(0, 512) ECFR_UP(4, 512) ECFR_UP(2, 512) }
        else { ECFR_UP(0) ECFR_UP(4) ECFR_UP(2) }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (store_all_f < 0 || store_all_f > 2) return fail(MGRIT_HIP_EINVAL, "store_all_f %d outside 0..2", store_all_f);
    if (!out_caller && (rc = ensure_pinned(e, lf.ivals_n[ivals_id]))) return rc;
    double *out = out_caller ? out_caller : e->pinned;
    Timed timed(e, MGRIT_HIP_T_EC_RELAX_RES, lvl);
    double *const *mirror = e->mirror_cur;   // null until mgrit_hip_cpoint_mirror has been called
    const int row0 = e->mirror_row0;
    const int fme = force_mode(lf) == 0 ? 0 : force_mode(lf) == 1 ? 4 : 2;
    const int tb = sweep_tb(lf.dev.T);
#define ECFR_RES(F, ...)                                                                                                              \
    if (fme == F) hipLaunchKernelGGL((ecfr_kernel<F, false, true, ##__VA_ARGS__>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, \
                                     sched_dev(e, lf), lc.dev, I, out, store_all_f, mirror, row0);
    if (tb == LANES) { ECFR_RES(0, LANES) ECFR_RES(4, LANES) ECFR_RES(2, LANES) }
    else if (tb == 512) { ECFR_RES(0, 512) ECFR_RES(4, 512) ECFR_RES(2, 512) }
    else { ECFR_RES(0) ECFR_RES(4) ECFR_RES(2) }
    HIP_TRY(hipGetLastError());
    return 0;
}

static int gen_check(mgrit_hip_engine *e, int lvl, int ivals_id, const char *what) {
    int rc = check_level(e, lvl);
    if (rc) return rc;
    if (lvl + 1 >= e->n_levels || !e->L[lvl + 1].set) return fail(MGRIT_HIP_EINVAL, "level %d has no coarser level", lvl);
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    if (lf.h2d || lc.h2d || is_2pts(lf) || is_2pts(lc) || lf.dev.kind != lc.dev.kind)
        return fail(MGRIT_HIP_EUNSUPPORTED, "%s needs the same 1-D single-point stepper on both levels", what);
    if ((rc = no_wide(lf, &lc, what))) return rc;
    if (ivals_id < 0 || ivals_id >= (int)lf.ivals.size()) return fail(MGRIT_HIP_EINVAL, "bad interval-list id %d on level %d", ivals_id, lvl);
    if ((rc = check_bound(lf, lvl > 0)) || (rc = check_bound(lc, true))) return rc;
    const int tk = lf.transfer;
    const bool fits = tk == MGRIT_HIP_TRANSFER_COPY ? lf.dev.n == lc.dev.n : tk == MGRIT_HIP_TRANSFER_HEAT1D ? lf.dev.n == 2 * lc.dev.n + 1
                    : tk == MGRIT_HIP_TRANSFER_PERIODIC1D ? lf.dev.n == 2 * lc.dev.n : false;
    if (!fits) return fail(MGRIT_HIP_EUNSUPPORTED, "%s: transfer kind %d does not join n=%d and n=%d", what, tk, lf.dev.n, lc.dev.n);
    return 0;
}

static int gen_reserve(mgrit_hip_engine *e, Level &lf, size_t res_len) {
    if (lf.gen_rows && lf.gen_len >= res_len) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(e->stream, &cs);
    if (cs != hipStreamCaptureStatusNone) return fail(MGRIT_HIP_EINVAL, "first whole-level pass of a level inside a stream capture");
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, res_len * (size_t)lf.dev.ld * sizeof(double)));
    lf.allocs.push_back(d);      // (an earlier, smaller slab stays in the list until the engine goes)
    lf.gen_rows = static_cast<double *>(d);
    lf.gen_len = res_len;
    return 0;
}

int mgrit_hip_gen_down(mgrit_hip_engine *e, int lvl, int ivals_id) { return mgrit_hip_gen_down_part(e, lvl, ivals_id, 3); }

int mgrit_hip_gen_down_part(mgrit_hip_engine *e, int lvl, int ivals_id, int parts) {
    int rc = gen_check(e, lvl, ivals_id, "whole-level way down");
    if (rc) return rc;
    if (parts < 1 || parts > 3) return fail(MGRIT_HIP_EINVAL, "parts %d: 1 = the fine level's pass, 2 = the coarse half, 3 = both", parts);
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    const IntervalsDev &I = lf.ivals[ivals_id];
    const int n_iv = lf.ivals_cnt[ivals_id];
    if (I.n_chunks == 0) return 0;
    const size_t res_len = (size_t)lf.ivals_n[ivals_id];
    if ((rc = gen_reserve(e, lf, res_len))) return rc;
    double *Cb = lf.gen_rows;
    const int tk = lf.transfer;
    Timed timed(e, MGRIT_HIP_T_GEN_DOWN, lvl);
    if (parts & 1) {
    const dim3 grid(persistent_grid(lf, I.n_chunks)), block(lf.dev.T);
    const bool use_g = lvl > 0;
    const int fm = force_mode(lf);
    const bool half = lf.dev.T <= 512 && gen_half_instances();   // (states of <= 8192 values: the instances compiled for 512 threads)
#define GEN_DOWN_CASE(K, F, G_)                                                                                     \
    if (lf.dev.kind == K && fm == F && use_g == G_) {                                                               \
        if (half) hipLaunchKernelGGL((gen_down_kernel<K, F, G_, 512>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), lc.dev, I, Cb, tk); \
        else hipLaunchKernelGGL((gen_down_kernel<K, F, G_>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), lc.dev, I, Cb, tk); \
    }
#define GEN_DOWN_CASES(K, F) GEN_DOWN_CASE(K, F, false) GEN_DOWN_CASE(K, F, true)
    FOR_EACH_STEPPER(GEN_DOWN_CASES)
    HIP_TRY(hipGetLastError());
    }
    if (parts & 2) { LAUNCH_BY_KIND(fas_coarse_kernel, lc, n_iv, lc.dev, I.cend_coarse, 1); }
    return 0;
}

int mgrit_hip_gen_up(mgrit_hip_engine *e, int lvl, int ivals_id, int with_residual, double *sumsq_out) {
    int rc = gen_check(e, lvl, ivals_id, "whole-level way up");
    if (rc) return rc;
    if (with_residual && lvl != 0) return fail(MGRIT_HIP_EINVAL, "the residual check belongs to level 0");
    Level &lf = e->L[lvl], &lc = e->L[lvl + 1];
    const IntervalsDev &I = lf.ivals[ivals_id];
    if (I.n_chunks == 0) return 0;
    const size_t res_len = (size_t)lf.ivals_n[ivals_id];
    if (!lf.gen_rows || lf.gen_len < res_len) return fail(MGRIT_HIP_EINVAL, "mgrit_hip_gen_up before the mgrit_hip_gen_down of the cycle");
    const double *Cb = lf.gen_rows;
    double *out = nullptr;
    if (with_residual) {
        if (!sumsq_out && (rc = ensure_pinned(e, (int)res_len))) return rc;
        out = sumsq_out ? sumsq_out : e->pinned;
    }
    Timed timed(e, MGRIT_HIP_T_GEN_UP, lvl);
    const dim3 grid(persistent_grid(lf, I.n_chunks)), block(lf.dev.T);
    const bool use_g = lvl > 0, res = with_residual != 0;
    const int fm = force_mode(lf), tk = lf.transfer;
    const bool half = lf.dev.T <= 512 && gen_half_instances();
#define GEN_UP_CASE(K, F, G_, R_)                                                                                    \
    if (lf.dev.kind == K && fm == F && use_g == G_ && res == R_) {                                                    \
        if (half) hipLaunchKernelGGL((gen_up_kernel<K, F, G_, R_, 512>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), lc.dev, I, Cb, out, tk); \
        else hipLaunchKernelGGL((gen_up_kernel<K, F, G_, R_>), grid, block, smem_bytes(lf.G, lf.dev.kind), e->stream, sched_dev(e, lf), lc.dev, I, Cb, out, tk); \
    }
#define GEN_UP_CASES(K, F) GEN_UP_CASE(K, F, false, false) GEN_UP_CASE(K, F, false, true) GEN_UP_CASE(K, F, true, false)
    FOR_EACH_STEPPER(GEN_UP_CASES)
    HIP_TRY(hipGetLastError());
    return 0;
}

int mgrit_hip_residual_fetch(mgrit_hip_engine *e, int n, double *sumsq_host) {
    if (!e || n < 0 || (n > 0 && !sumsq_host)) return fail(MGRIT_HIP_EINVAL, "bad arguments");
    if ((size_t)n > e->pinned_len) return fail(MGRIT_HIP_EINVAL, "no %d residual values have been produced", n);
    if (n == 0) return 0;
    return wait_pinned(e, n, sumsq_host);
}

int mgrit_hip_set_reserve(mgrit_hip_engine *e, int n_cus) {
    if (!e) return fail(MGRIT_HIP_EINVAL, "null engine");
    if (n_cus < 0 || n_cus > 32) return fail(MGRIT_HIP_EINVAL, "reserve %d outside [0,32]", n_cus);
    int rc;
    if (n_cus > 0 && (rc = ensure_sched(e))) return rc;
    e->reserve = n_cus;
    return 0;
}

int mgrit_hip_chain_clock(mgrit_hip_engine *e, double *mhz, double *us_per_step) {
    if (!e || !mhz || !us_per_step) return fail(MGRIT_HIP_EINVAL, "null argument");
    *mhz = *us_per_step = 0.0;
    if (!e->chain_err) return 0;
    const volatile unsigned long long *w = reinterpret_cast<const volatile unsigned long long *>(e->chain_err);
    if (w[2] == 0 || w[3] == 0) return 0;
    *mhz = 100.0 * (double)w[1] / (double)w[2];
    *us_per_step = (double)w[2] / 100.0 / (double)w[3];
    return 0;
}

int mgrit_hip_set_stream(mgrit_hip_engine *e, void *stream) {
    if (!e) return fail(MGRIT_HIP_EINVAL, "null engine");
    e->stream = static_cast<hipStream_t>(stream);
    return 0;
}

}  // extern "C"

#include "mgrit_hip_comm.inc"
