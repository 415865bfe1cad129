// K1 `linearize` and K2 `propagate`: the per-segment discretisation of the SCvx inner loop on gfx950.
//
// Replaces Dynamics.linearize_dynamics (dynamics.jl:321-334 -> sensitivity :298-305) and
// Dynamics.predict_state (dynamics.jl:315-317 -> simulate :288-296) of the reference, using the
// fixed-step RK4 it names in Dynamics.rk4 (dynamics.jl:112-134, `npts` substeps, FOH control at
// substep start / middle / end) without that routine's stage bug (:126-128 drop the step size).
//
// K1 work decomposition (wave64; aero instantiation — the exo one drops six closed-form columns, see K1Map):
// the 21 columns of the 14x21 sensitivity
//     d x(t) / d [x_k | u_k | u_{k+1} | sigma]
// obey independent linear ODEs once the state trajectory is known, so ONE LANE OWNS ONE COLUMN and
// keeps it, its RK4 accumulator and its stage value in VGPRs (no LDS traffic, no cross-lane
// dependency inside the integration).  A wavefront carries three consecutive segments
// (3 x 21 = 63 lanes; lane 63 idles); every lane integrates its segment's 14 states redundantly,
// which is cheaper than broadcasting them.  df/dx is never materialised: each lane applies the ~48
// structural non-zeros straight to its column (scvx_dyn.hpp).  The finished 3 x 14 x 21 tile is
// transposed through LDS so that the wavefront writes its 7,056 contiguous output bytes with
// 16-byte-per-lane coalesced stores in the reference's column-major LinRes layout.
#include <type_traits>
#include "scvx_internal.hpp"

namespace scvx {

constexpr int WAVES_PER_BLOCK = 4;

// Lanes per segment and segments per wavefront.
//   aero : 21 lanes, one per column, 3 segments per wave (63 lanes).
//   exo  : 15 lanes, 4 segments per wave (60 lanes).  Without aerodynamics nothing depends on position or velocity
//          except r' = sigma v, so six columns are closed form under RK4 (exact for polynomials in t):
//            d x(dt)/d r_k = [0; I; 0; 0; 0],   d x(dt)/d v_k = [0; sigma dt I; I; 0; 0]
//          and only the columns of m, q(4), w(3), u_k(3), u_{k+1}(3), sigma are integrated.
//   fins : 25 lanes (control_dim = 5: 14 + 5 + 5 + 1 columns), 2 segments per wave (50 lanes); the fin force depends on the
//          velocity, so no column is closed form.
template <bool AERO, bool FIN = false> struct K1Map {
    static constexpr int LPS = FIN ? 25 : (AERO ? 21 : 15), SPW = FIN ? 2 : (AERO ? 3 : 4);
    static constexpr int NU = FIN ? 5 : 3, NP = 14 + 2 * NU + 1, DSZ = 14 * NP, HV = DSZ / 2;   // tile: DSZ values = HV 16-byte pairs
};
__device__ __forceinline__ int exo_slot_to_col(int slot) {  // 0 -> m, 1..7 -> q,w, 8..14 -> u_k,u_{k+1},sigma
    return slot == 0 ? 0 : slot + 6;
}

// solve_step re-linearises after the trust-region test, but a REJECTED step keeps about / dynam (rocketland.jl:299-301) and
// a frozen trajectory never changes: `skip` = the per-trajectory step status (>= SCVX_ST_REJECTED: unchanged iterate).  A
// block whose segments all belong to such trajectories returns at once (uniform for the block: before any barrier).
__device__ __forceinline__ bool block_unchanged(const int* skip, long seg0, int nseg_block, long nseg, int K) {
    if (!skip || seg0 >= nseg) return false;
    long seg1 = seg0 + nseg_block - 1;
    if (seg1 >= nseg) seg1 = nseg - 1;
    const long b0 = seg0 / K, b1 = seg1 / K;
    for (long b = b0; b <= b1; b++)
        if (skip[b] < SCVX_ST_REJECTED) return false;
    return true;
}

template <typename R> struct Vec2;
template <> struct Vec2<double> { typedef double2 type; };
template <> struct Vec2<float> { typedef float2 type; };

// float: 2 wavefronts per SIMD (the four unrolled stages keep ~250 values live either way: at 168 VGPRs 300-600 spill)
template <bool AERO, typename R>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, sizeof(R) == 4 ? 2 : 1) void linearize_kernel(
    DynP<R> p, long nseg, int K, const R* __restrict__ x, const R* __restrict__ u,
    const R* __restrict__ sigma, R dt, int nsub, R* __restrict__ endpoint,
    R* __restrict__ deriv, const int* __restrict__ skip) {
    constexpr int LPS = K1Map<AERO>::LPS, SPW = K1Map<AERO>::SPW;
    typedef typename Vec2<R>::type VEC2;   // 16-byte (double) / 8-byte (float) pairs: 147 per segment either way
    if (block_unchanged(skip, (long)blockIdx.x * WAVES_PER_BLOCK * SPW, WAVES_PER_BLOCK * SPW, nseg, K)) return;
    __shared__ __attribute__((aligned(16))) R tile[WAVES_PER_BLOCK][SPW * 294];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long gwave = (long)blockIdx.x * WAVES_PER_BLOCK + wave;
    const int sl = lane / LPS;
    const int slot = lane - sl * LPS;
    const int col = AERO ? slot : exo_slot_to_col(slot);
    const long seg0 = gwave * SPW;
    long seg = seg0 + sl;
    const bool active = (sl < SPW) && (seg < nseg);
    if (!active) seg = (seg0 < nseg) ? seg0 : nseg - 1;  // idle lanes shadow a valid segment, never store
    const long b = seg / K;
    const int k = (int)(seg - b * K);
    const R* xk = x + ((size_t)b * (K + 1) + k) * 14;
    const R* uk = u + ((size_t)b * (K + 1) + k) * 3;
    const R sig = sigma[b];

    R xs[14], c[14];
#pragma unroll
    for (int i = 0; i < 14; i++) {
        xs[i] = xk[i];
        c[i] = (col == i) ? R(1.0) : R(0.0);
    }
    const R uk0 = uk[0], uk1 = uk[1], uk2 = uk[2];
    const R up0 = uk[3], up1 = uk[4], up2 = uk[5];
    const bool is_uk = (col >= 14) && (col < 17);
    const bool is_up = (col >= 17) && (col < 20);
    const int comp = is_uk ? col - 14 : (is_up ? col - 17 : -1);
    const R gsel = (col == 20) ? R(1.0) : R(0.0);
    const R e0 = (comp == 0) ? R(1.0) : R(0.0), e1 = (comp == 1) ? R(1.0) : R(0.0), e2 = (comp == 2) ? R(1.0) : R(0.0);

    const R h = dt / R(nsub);
    const R inv_n = R(1.0) / R(nsub);
    for (int s = 0; s < nsub; s++) {
        R xa[14], ca[14], xt[14], ct[14];
#pragma unroll
        for (int i = 0; i < 14; i++) {
            xa[i] = xs[i];
            ca[i] = c[i];
            xt[i] = xs[i];
            ct[i] = c[i];
        }
#pragma unroll
        for (int stg = 0; stg < 4; stg++) {
            const R lkp = (R(s) + (stg == 0 ? R(0.0) : (stg == 3 ? R(1.0) : R(0.5)))) * inv_n;
            const R lkm = R(1.0) - lkp;
            R uu[3] = {fma(uk0, lkm, up0 * lkp), fma(uk1, lkm, up1 * lkp), fma(uk2, lkm, up2 * lkp)};
            const R wk = is_uk ? lkm : (is_up ? lkp : R(0.0));
            const R wc[3] = {e0 * wk, e1 * wk, e2 * wk};
            Stage<AERO, R> st;
            stage_eval<AERO>(p, xt, uu, st);
            R dc[14];
            column_deriv<AERO>(p, st, xt, uu, ct, wc, gsel, sig, dc);
            const R wacc = h * ((stg == 0 || stg == 3) ? (R(1.0) / R(6.0)) : (R(1.0) / R(3.0)));
            const R wnext = h * (stg == 2 ? R(1.0) : R(0.5));
#pragma unroll
            for (int i = 0; i < 14; i++) {
                const R dx = sig * st.g[i];
                xa[i] = fma(wacc, dx, xa[i]);
                ca[i] = fma(wacc, dc[i], ca[i]);
                if (stg < 3) {
                    xt[i] = fma(wnext, dx, xs[i]);
                    ct[i] = fma(wnext, dc[i], c[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 14; i++) {
            xs[i] = xa[i];
            c[i] = ca[i];
        }
    }

    // ---- epilogue: columns into the LDS tile -> coalesced 16-byte stores of the wave's contiguous tile ----
    R* t = tile[wave];
    if (sl < SPW) {
#pragma unroll
        for (int i = 0; i < 14; i++) t[sl * 294 + col * 14 + i] = c[i];
        if (!AERO && slot < 6) {  // closed-form columns: slots 0..2 write d/dr_k, slots 3..5 write d/dv_k
            const int j = slot < 3 ? slot : slot - 3;
            R* cc = t + sl * 294 + (slot < 3 ? 1 + j : 4 + j) * 14;
#pragma unroll
            for (int i = 0; i < 14; i++) cc[i] = R(0.0);
            if (slot < 3) cc[1 + j] = R(1.0);
            else { cc[1 + j] = sig * dt; cc[4 + j] = R(1.0); }
        }
    }
    __syncthreads();
    if (seg0 < nseg) {
        const long rem = nseg - seg0;
        const int nvalid = rem < SPW ? (int)rem : SPW;
        const int n2 = nvalid * 147;  // VEC2 elements in the tile (294 / 2 per segment)
        VEC2* out = reinterpret_cast<VEC2*>(deriv + (size_t)seg0 * 294);
        const VEC2* src = reinterpret_cast<const VEC2*>(t);
#pragma unroll
        for (int r = 0; r < (SPW * 147 + 63) / 64; r++) {
            const int e = lane + 64 * r;
            if (e < n2) out[e] = src[e];
        }
        if (active && slot == 0) {
            R* ep = endpoint + (size_t)seg * 14;
#pragma unroll
            for (int i = 0; i < 14; i++) ep[i] = xs[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// K1, producer / consumer form.  In linearize_kernel every lane repeats the 14-state stage evaluation (~250 of the
// ~440 instructions per stage) because SIMD lanes of a segment run in lock-step.  Here a block of 8 wavefronts
// splits the roles:
//   wave 0   (producer)  one LANE PER SEGMENT: integrates the 14 states of NS segments and publishes, for every RK
//                        stage, the coefficient record a column needs (StageRec) into a double-buffered LDS slab --
//                        group by group as it is computed, the force derivatives one column at a time
//                        (stage_eval_publish, scvx_dyn.hpp: only the RHS stays live for the state update);
//   waves 1-7 (consumers) one lane per sensitivity column (15 per segment exo, 4 segments per wave; 21 x 3 aero; 25 x 2 with
//                        the fin extension): read their segment's record in three batches (LDS broadcast), advance the column.
// One barrier per RK stage, the producer one stage ahead through a 2-slot ring (SG; or per substep, one substep ahead,
// 8 slots).  The stage evaluation is executed once per 28 (21)
// segments instead of once per 4 (3): ~1.8x fewer instructions per segment.  Output tiles leave through LDS as
// coalesced 16-byte stores, exactly as in linearize_kernel.
// ------------------------------------------------------------------------------------------------------------
// Block shape (measured at B = 8192, npts 10; -DSCVX_PC_WAVES / -DSCVX_PC_PROD rebuild the variants): 8 wavefronts with
// the producer first is the best that fits — 2.96 ms; 7 / 6 / 5 wavefronts 3.29 / 3.60 / 4.28 ms (the time per substep of
// a block does not depend on its width: each wavefront is latency-bound, so throughput goes with the number of consumer
// wavefronts per CU); 10 or 12 wavefronts need <= 168 VGPRs, spill 260 registers and run 3-4x slower.
#ifndef SCVX_PC_WAVES
#define SCVX_PC_WAVES 8
#endif
#ifndef SCVX_PC_PROD
#define SCVX_PC_PROD 0
#endif
#ifndef SCVX_PC_BLOCKS_PER_CU
#define SCVX_PC_BLOCKS_PER_CU 1
#endif
constexpr int PC_BLOCKS_PER_CU = SCVX_PC_BLOCKS_PER_CU;
constexpr int PC_WAVES = SCVX_PC_WAVES;
constexpr int PC_PROD = SCVX_PC_PROD;
// workgroup barrier that orders LDS traffic only: global loads and stores stay in flight across it (__syncthreads() also
// drains vmcnt)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#if defined(SCVX_K1_PROF)
// diagnostic build: per-wavefront cycles spent waiting at the stage barriers of the persistent kernel (block 0), read back by
// scvx_debug_k1_prof: [wave][0] = cycles in barriers, [wave][1] = cycles in the kernel
__device__ double g_k1prof[2 * 16];
#define K1_BAR() do { const unsigned long long tb_ = __builtin_amdgcn_s_memtime(); lds_barrier(); k1wait_ += (double)(__builtin_amdgcn_s_memtime() - tb_); } while (0)
#else
#define K1_BAR() lds_barrier()
#endif
template <bool AERO, bool FIN, typename R>
__device__ __forceinline__ void column_deriv_rec_any(const DynP<R>& p, const R* rec, int stride, const R* c, const R* wc, R gsel,
                                                     R sigma, R* dc) {
    column_deriv_rec_pieces<AERO, FIN, R>(p, rec, stride, c, wc, gsel, sigma, dc);
}
constexpr int PC_GROUP = 4;              // stages published per barrier (one RK4 substep)
// SG (stage-granular, the default): one barrier per RK stage and the producer one STAGE ahead (2-slot ring) instead of
// one barrier per substep and the producer one substep ahead: the pipeline fills after one stage instead of four (at
// npts = 1 the substep-granular form does not overlap at all), which outweighs the 4x barrier count at every npts.
// O = element type of the derivative tiles in HBM: R, or float under double arithmetic (scvx_batch_set_linearization_f32:
// the conic solve reads them as float; rounded once, at the store)
template <bool AERO, bool SG, typename R, typename O = R, bool FIN = false>
__global__ __launch_bounds__(64 * PC_WAVES) void linearize_pc_kernel(
    DynP<R> p, long nseg, int K, const R* __restrict__ x, const R* __restrict__ u,
    const R* __restrict__ sigma, R dt, int nsub, R* __restrict__ endpoint,
    O* __restrict__ deriv, const int* __restrict__ skip) {
    constexpr int LPS = K1Map<AERO, FIN>::LPS, SPW = K1Map<AERO, FIN>::SPW;
    constexpr int NU = K1Map<AERO, FIN>::NU, NP = K1Map<AERO, FIN>::NP, DSZ = K1Map<AERO, FIN>::DSZ, HV = K1Map<AERO, FIN>::HV;
    typedef typename Vec2<R>::type VEC2;
    typedef typename Vec2<O>::type OVEC2;
    constexpr int NC = PC_WAVES - 1;
    if (block_unchanged(skip, (long)blockIdx.x * (NC * SPW), NC * SPW, nseg, K)) return;
    constexpr int NS = NC * SPW;               // segments per block
    constexpr int NR = StageRec<AERO, FIN>::N;
    constexpr int RING = SG ? 2 : 2 * PC_GROUP;   // stage records in flight: the producer runs one stage / one substep ahead
    constexpr int RING_D = RING * NR * NS, TILE_D = NC * SPW * DSZ;
    // one LDS slab: the coefficient ring during the integration, the output tiles afterwards
    __shared__ __attribute__((aligned(16))) R lds[RING_D > TILE_D ? RING_D : TILE_D];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long seg_base = (long)blockIdx.x * NS;
    const R h = dt / R(nsub);
    const R inv_n = R(1.0) / R(nsub);

    if (wave == PC_PROD) {
        // ---------------- producer: lane = segment ----------------
        const bool live = lane < NS;
        long seg = seg_base + (live ? lane : 0);
        const bool valid = live && seg < nseg;
        if (seg >= nseg) seg = nseg - 1;
        const long b = seg / K;
        const int k = (int)(seg - b * K);
        const R* xk = x + ((size_t)b * (K + 1) + k) * 14;
        const R* uk = u + ((size_t)b * (K + 1) + k) * NU;
        const R sig = sigma[b];
        R xs[14], xa[14], xt[14];
#pragma unroll
        for (int i = 0; i < 14; i++) { xs[i] = xk[i]; xa[i] = xs[i]; xt[i] = xs[i]; }
        R ukv[NU], upv[NU];
#pragma unroll
        for (int j = 0; j < NU; j++) { ukv[j] = uk[j]; upv[j] = uk[NU + j]; }
        const int l = live ? lane : 0;
        for (int s = 0; s <= nsub; s++) {
            if (s < nsub) {
#pragma unroll
                for (int stg = 0; stg < 4; stg++) {
                    const R lkp = (R(s) + (stg == 0 ? R(0.0) : (stg == 3 ? R(1.0) : R(0.5)))) * inv_n;
                    const R lkm = R(1.0) - lkp;
                    R uu[NU];
#pragma unroll
                    for (int j = 0; j < NU; j++) uu[j] = fma(ukv[j], lkm, upv[j] * lkp);
                    struct { R g[14]; } st;
                    stage_eval_publish<AERO, FIN>(p, xt, uu, st.g, lds + (SG ? (stg & 1) : (s & 1) * PC_GROUP + stg) * NR * NS + l, NS, live);
                    const R wacc = h * ((stg == 0 || stg == 3) ? (R(1.0) / R(6.0)) : (R(1.0) / R(3.0)));
                    const R wnext = h * (stg == 2 ? R(1.0) : R(0.5));
#pragma unroll
                    for (int i = 0; i < 14; i++) {
                        const R dx = sig * st.g[i];
                        xa[i] = fma(wacc, dx, xa[i]);
                        xt[i] = (stg < 3) ? fma(wnext, dx, xs[i]) : xa[i];
                    }
                    if (SG) __syncthreads();   // stage 4 s + stg is published
                }
#pragma unroll
                for (int i = 0; i < 14; i++) xs[i] = xa[i];
            }
            if (!SG || s == nsub) __syncthreads();
        }
        if (valid) {
            R* ep = endpoint + (size_t)seg * 14;
#pragma unroll
            for (int i = 0; i < 14; i++) ep[i] = xs[i];
        }
        __syncthreads();  // matches the consumers' tile barrier
        return;
    }

    // ---------------- consumers: lane = (segment, column) ----------------
    const int cw = wave < PC_PROD ? wave : wave - 1;
    const int sl = lane / LPS;
    const int slot = lane - sl * LPS;
    const int col = (AERO || FIN) ? slot : exo_slot_to_col(slot);
    const bool lane_live = sl < SPW;
    const int ls = cw * SPW + (lane_live ? sl : 0);   // local segment index in the block
    long seg = seg_base + ls;
    if (seg >= nseg) seg = nseg - 1;
    const R sig = sigma[seg / K];
    R c[14], ca[14], ct[14];
#pragma unroll
    for (int i = 0; i < 14; i++) { c[i] = (col == i) ? R(1.0) : R(0.0); ca[i] = c[i]; ct[i] = c[i]; }
    const bool is_uk = (col >= 14) && (col < 14 + NU);
    const bool is_up = (col >= 14 + NU) && (col < 14 + 2 * NU);
    const int comp = is_uk ? col - 14 : (is_up ? col - 14 - NU : -1);
    const R gsel = (col == NP - 1) ? R(1.0) : R(0.0);
    R ec[NU];
#pragma unroll
    for (int j = 0; j < NU; j++) ec[j] = (comp == j) ? R(1.0) : R(0.0);
    __syncthreads();  // records of substep 0 are ready
    for (int s = 0; s < nsub; s++) {
#pragma unroll
        for (int stg = 0; stg < 4; stg++) {
            const R lkp = (R(s) + (stg == 0 ? R(0.0) : (stg == 3 ? R(1.0) : R(0.5)))) * inv_n;
            const R lkm = R(1.0) - lkp;
            const R wk = is_uk ? lkm : (is_up ? lkp : R(0.0));
            R wc[NU];
#pragma unroll
            for (int j = 0; j < NU; j++) wc[j] = ec[j] * wk;
            R dc[14];
            column_deriv_rec_any<AERO, FIN>(p, lds + (SG ? (stg & 1) : (s & 1) * PC_GROUP + stg) * NR * NS + ls, NS, ct, wc, gsel, sig, dc);
            const R wacc = h * ((stg == 0 || stg == 3) ? (R(1.0) / R(6.0)) : (R(1.0) / R(3.0)));
            const R wnext = h * (stg == 2 ? R(1.0) : R(0.5));
#pragma unroll
            for (int i = 0; i < 14; i++) {
                ca[i] = fma(wacc, dc[i], ca[i]);
                ct[i] = (stg < 3) ? fma(wnext, dc[i], c[i]) : ca[i];
            }
            if (SG) __syncthreads();   // this slot is free again / the next stage is published
        }
#pragma unroll
        for (int i = 0; i < 14; i++) c[i] = ca[i];
        if (!SG) __syncthreads();
    }
    // ---- epilogue: columns into the LDS tile (the ring is dead now) -> coalesced 16-byte stores ----
    R* t = lds + cw * SPW * DSZ;
    if (lane_live) {
#pragma unroll
        for (int i = 0; i < 14; i++) t[sl * DSZ + col * 14 + i] = c[i];
        if (!AERO && !FIN && slot < 6) {
            const int j = slot < 3 ? slot : slot - 3;
            R* cc = t + sl * DSZ + (slot < 3 ? 1 + j : 4 + j) * 14;
#pragma unroll
            for (int i = 0; i < 14; i++) cc[i] = R(0.0);
            if (slot < 3) cc[1 + j] = R(1.0);
            else { cc[1 + j] = sig * dt; cc[4 + j] = R(1.0); }
        }
    }
    __syncthreads();
    const long seg0 = seg_base + (long)cw * SPW;
    if (seg0 < nseg) {
        const long rem = nseg - seg0;
        const int nvalid = rem < SPW ? (int)rem : SPW;
        const int n2 = nvalid * HV;
        OVEC2* out = reinterpret_cast<OVEC2*>(deriv + (size_t)seg0 * DSZ);
        const VEC2* src = reinterpret_cast<const VEC2*>(t);
#pragma unroll
        for (int r = 0; r < (SPW * HV + 63) / 64; r++) {
            const int e = lane + 64 * r;
            if (e < n2) { const VEC2 v = src[e]; OVEC2 o; o.x = O(v.x); o.y = O(v.y); out[e] = o; }
        }
    }
}

// The same producer/consumer pipeline (stage-granular) as a PERSISTENT block, used from 3 substeps up.
// NB (round 6): segment batches per consumer lane, as in the split kernel below -- each consumer lane carries its sensitivity column of NB
// segments (NB sets of c / accumulator / stage value), the producer's lanes cover NB times the segments (a lane per segment: 56 of 64 lanes
// instead of 28 without aerodynamics) and a tick serves NB x 28 segments.  SCVX_K1_NB_EXO selects it for the model without aerodynamics.
#ifndef SCVX_K1_NB_EXO
#define SCVX_K1_NB_EXO 1   // MEASURED AND OFF: 2 = 2.81 -> 6.16 ms at npts 10 (200 spilled VGPRs: profiles/r06_k1_exo_nb2.md)
#endif
template <bool AERO, typename R, typename O = R, bool FIN = false, int NB = 1>
__global__ __launch_bounds__(64 * PC_WAVES) void linearize_pcp_kernel(
    DynP<R> p, long nseg, int K, const R* __restrict__ x, const R* __restrict__ u,
    const R* __restrict__ sigma, R dt, int nsub, R* __restrict__ endpoint,
    O* __restrict__ deriv, const int* __restrict__ skip) {
    constexpr int LPS = K1Map<AERO, FIN>::LPS, SPW = K1Map<AERO, FIN>::SPW;
    constexpr int NU = K1Map<AERO, FIN>::NU, NP = K1Map<AERO, FIN>::NP, DSZ = K1Map<AERO, FIN>::DSZ, HV = K1Map<AERO, FIN>::HV;
    typedef typename Vec2<R>::type VEC2;
    typedef typename Vec2<O>::type OVEC2;
    constexpr bool SG = true;
    constexpr int NC = PC_WAVES - 1;
    constexpr int NS = NC * SPW * NB;          // segments per group
    static_assert(NS <= 64, "one producer lane per segment");
    constexpr int NR = StageRec<AERO, FIN>::N;
    constexpr int RING = SG ? 2 : 2 * PC_GROUP;   // stage records in flight: the producer runs one stage / one substep ahead
    constexpr int RING_D = RING * NR * NS, TILE_D = NC * SPW * DSZ;
    // PERSISTENT block: it walks the groups of NS segments blockIdx.x, blockIdx.x + gridDim.x, ...  The output tiles of a
    // group leave through 16-byte global stores that nothing waits for: every barrier in here orders LDS only
    // (lds_barrier), so the 2.4 KB per segment of group g drain to HBM while group g + 1 integrates -- with one block per
    // CU and a block per group, the store phase (0.3 ms of a launch at B = 8192, the whole intercept of the time-vs-npts
    // line) ran after the arithmetic instead of under it.  Ring and tiles are separate LDS regions for that reason.
    constexpr bool SHARE = false;
    // (the one-group-per-block kernel above keeps ring and tiles in one 66 KB slab, so a CU's LDS holds a second block whose
    // wavefronts start as the first block's retire: the better shape when a group is latency-bound, npts <= 2 -- measured)
    __shared__ __attribute__((aligned(16))) R lds[SHARE ? (RING_D > TILE_D ? RING_D : TILE_D) : RING_D + TILE_D];
    R* const tiles = SHARE ? lds : lds + RING_D;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
#if defined(SCVX_K1_PROF)
    double k1wait_ = 0.0;
    const unsigned long long k1t0_ = __builtin_amdgcn_s_memtime();
#endif
    const R h = dt / R(nsub);
    const R inv_n = R(1.0) / R(nsub);
    const long ngrp = (nseg + NS - 1) / NS;
    // next group of this block that has something to recompute (uniform for the block)
    auto advance = [&](long g) {
        while (g < ngrp && block_unchanged(skip, g * NS, NS, nseg, K)) g += gridDim.x;
        return g;
    };
    long grp = advance(blockIdx.x);
    // vmcnt counts loads and stores in one in-order queue: a global load issued after a group's stores could only be
    // waited for together with them.  So each role fetches the inputs of its NEXT group before it issues the stores of
    // the current one, and nothing waits on vmcnt between a group's stores and the end of the next group's arithmetic.

    if (wave == PC_PROD) {
        // ---------------- producer: lane = segment ----------------
        const bool live = lane < NS;
        const int l = live ? lane : 0;
        R nx[14], nu6[2 * NU], nsig = R(0.0);
        auto fetch = [&](long g) {
            long seg = g * NS + l;
            if (seg >= nseg) seg = nseg - 1;
            const long b = seg / K;
            const int k = (int)(seg - b * K);
            const R* xk = x + ((size_t)b * (K + 1) + k) * 14;
            const R* uk = u + ((size_t)b * (K + 1) + k) * NU;
#pragma unroll
            for (int i = 0; i < 14; i++) nx[i] = xk[i];
#pragma unroll
            for (int i = 0; i < 2 * NU; i++) nu6[i] = uk[i];
            nsig = sigma[b];
        };
        if (grp < ngrp) fetch(grp);
        while (grp < ngrp) {
            const long nxt = advance(grp + gridDim.x);
            const long seg = grp * NS + l;
            const bool valid = live && seg < nseg;
            const R sig = nsig;
            R xs[14], xa[14], xt[14];
#pragma unroll
            for (int i = 0; i < 14; i++) { xs[i] = nx[i]; xa[i] = xs[i]; xt[i] = xs[i]; }
            R ukv[NU], upv[NU];
#pragma unroll
            for (int j = 0; j < NU; j++) { ukv[j] = nu6[j]; upv[j] = nu6[NU + j]; }
            for (int s = 0; s <= nsub; s++) {
                if (s < nsub) {
#pragma unroll
                    for (int stg = 0; stg < 4; stg++) {
                        const R lkp = (R(s) + (stg == 0 ? R(0.0) : (stg == 3 ? R(1.0) : R(0.5)))) * inv_n;
                        const R lkm = R(1.0) - lkp;
                        R uu[NU];
#pragma unroll
                        for (int j = 0; j < NU; j++) uu[j] = fma(ukv[j], lkm, upv[j] * lkp);
                        struct { R g[14]; } st;
                        stage_eval_publish<AERO, FIN>(p, xt, uu, st.g, lds + (SG ? (stg & 1) : (s & 1) * PC_GROUP + stg) * NR * NS + l, NS, live);
                        const R wacc = h * ((stg == 0 || stg == 3) ? (R(1.0) / R(6.0)) : (R(1.0) / R(3.0)));
                        const R wnext = h * (stg == 2 ? R(1.0) : R(0.5));
#pragma unroll
                        for (int i = 0; i < 14; i++) {
                            const R dx = sig * st.g[i];
                            xa[i] = fma(wacc, dx, xa[i]);
                            xt[i] = (stg < 3) ? fma(wnext, dx, xs[i]) : xa[i];
                        }
                        if (SG) K1_BAR();   // stage 4 s + stg is published
                    }
#pragma unroll
                    for (int i = 0; i < 14; i++) xs[i] = xa[i];
                }
                if (!SG || s == nsub) K1_BAR();
            }
            if (nxt < ngrp) fetch(nxt);
            if (valid) {
                R* ep = endpoint + (size_t)seg * 14;
#pragma unroll
                for (int i = 0; i < 14; i++) ep[i] = xs[i];
            }
#pragma unroll
            for (int bb = 0; bb < NB; bb++) K1_BAR();  // matches the consumers' tile barriers
            if (SHARE) K1_BAR();   // the tiles have been read out of the shared slab
            grp = nxt;
        }
#if defined(SCVX_K1_PROF)
        if (blockIdx.x == 0 && lane == 0) { g_k1prof[2 * wave] = k1wait_; g_k1prof[2 * wave + 1] = (double)(__builtin_amdgcn_s_memtime() - k1t0_); }
#endif
        return;
    }

    // ---------------- consumers: lane = (segment, column) ----------------
    const int cw = wave < PC_PROD ? wave : wave - 1;
    const int sl = lane / LPS;
    const int slot = lane - sl * LPS;
    const int col = (AERO || FIN) ? slot : exo_slot_to_col(slot);
    const bool lane_live = sl < SPW;
    int ls[NB];                                        // local segment index in the group, per batch
#pragma unroll
    for (int bb = 0; bb < NB; bb++) ls[bb] = (bb * NC + cw) * SPW + (lane_live ? sl : 0);
    const bool is_uk = (col >= 14) && (col < 14 + NU);
    const bool is_up = (col >= 14 + NU) && (col < 14 + 2 * NU);
    const int comp = is_uk ? col - 14 : (is_up ? col - 14 - NU : -1);
    const R gsel = (col == NP - 1) ? R(1.0) : R(0.0);
    R ec[NU];
#pragma unroll
    for (int j = 0; j < NU; j++) ec[j] = (comp == j) ? R(1.0) : R(0.0);
    auto sigma_of = [&](long g, int bb) {
        long seg = g * NS + ls[bb];
        if (seg >= nseg) seg = nseg - 1;
        return sigma[seg / K];
    };
    R nsig[NB];
#pragma unroll
    for (int bb = 0; bb < NB; bb++) nsig[bb] = grp < ngrp ? sigma_of(grp, bb) : R(0.0);
    while (grp < ngrp) {
        const long nxt = advance(grp + gridDim.x);
        const long seg_base = grp * NS;
        R sig[NB];
        R c[NB][14], ca[NB][14], ct[NB][14];
#pragma unroll
        for (int bb = 0; bb < NB; bb++) {
            sig[bb] = nsig[bb];
#pragma unroll
            for (int i = 0; i < 14; i++) { c[bb][i] = (col == i) ? R(1.0) : R(0.0); ca[bb][i] = c[bb][i]; ct[bb][i] = c[bb][i]; }
        }
        K1_BAR();  // records of substep 0 are ready
        for (int s = 0; s < nsub; s++) {
#pragma unroll
            for (int stg = 0; stg < 4; stg++) {
                const R lkp = (R(s) + (stg == 0 ? R(0.0) : (stg == 3 ? R(1.0) : R(0.5)))) * inv_n;
                const R lkm = R(1.0) - lkp;
                const R wk = is_uk ? lkm : (is_up ? lkp : R(0.0));
                R wc[NU];
#pragma unroll
                for (int j = 0; j < NU; j++) wc[j] = ec[j] * wk;
                const R wacc = h * ((stg == 0 || stg == 3) ? (R(1.0) / R(6.0)) : (R(1.0) / R(3.0)));
                const R wnext = h * (stg == 2 ? R(1.0) : R(0.5));
#pragma unroll
                for (int bb = 0; bb < NB; bb++) {
                    R dc[14];
                    column_deriv_rec_any<AERO, FIN>(p, lds + (SG ? (stg & 1) : (s & 1) * PC_GROUP + stg) * NR * NS + ls[bb], NS, ct[bb], wc, gsel, sig[bb], dc);
#pragma unroll
                    for (int i = 0; i < 14; i++) {
                        ca[bb][i] = fma(wacc, dc[i], ca[bb][i]);
                        ct[bb][i] = (stg < 3) ? fma(wnext, dc[i], c[bb][i]) : ca[bb][i];
                    }
                }
                if (SG) K1_BAR();   // this slot is free again / the next stage is published
            }
#pragma unroll
            for (int bb = 0; bb < NB; bb++)
#pragma unroll
                for (int i = 0; i < 14; i++) c[bb][i] = ca[bb][i];
            if (!SG) K1_BAR();
        }
#pragma unroll
        for (int bb = 0; bb < NB; bb++) if (nxt < ngrp) nsig[bb] = sigma_of(nxt, bb);
        // ---- epilogue, batch by batch through this wavefront's LDS tile: columns in -> coalesced 16-byte stores out ----
        R* t = tiles + cw * SPW * DSZ;
#pragma unroll
        for (int bb = 0; bb < NB; bb++) {
            if (lane_live) {
#pragma unroll
                for (int i = 0; i < 14; i++) t[sl * DSZ + col * 14 + i] = c[bb][i];
                if (!AERO && !FIN && slot < 6) {
                    const int j = slot < 3 ? slot : slot - 3;
                    R* cc = t + sl * DSZ + (slot < 3 ? 1 + j : 4 + j) * 14;
#pragma unroll
                    for (int i = 0; i < 14; i++) cc[i] = R(0.0);
                    if (slot < 3) cc[1 + j] = R(1.0);
                    else { cc[1 + j] = sig[bb] * dt; cc[4 + j] = R(1.0); }
                }
            }
            K1_BAR();
            const long seg0 = seg_base + (long)(bb * NC + cw) * SPW;
            if (seg0 < nseg) {
                const long rem = nseg - seg0;
                const int nvalid = rem < SPW ? (int)rem : SPW;
                const int n2 = nvalid * HV;
                OVEC2* out = reinterpret_cast<OVEC2*>(deriv + (size_t)seg0 * DSZ);
                const VEC2* src = reinterpret_cast<const VEC2*>(t);
#pragma unroll
                for (int r = 0; r < (SPW * HV + 63) / 64; r++) {
                    const int e = lane + 64 * r;
                    if (e < n2) { const VEC2 v = src[e]; OVEC2 o; o.x = O(v.x); o.y = O(v.y); out[e] = o; }
                }
            }
        }
        if (SHARE) K1_BAR();
        grp = nxt;
    }
#if defined(SCVX_K1_PROF)
    if (blockIdx.x == 0 && lane == 0) { g_k1prof[2 * wave] = k1wait_; g_k1prof[2 * wave + 1] = (double)(__builtin_amdgcn_s_memtime() - k1t0_); }
#endif
}

// The persistent pipeline with the producer's stage split over TWO wavefronts (aero / fin models; scvx_dyn.hpp: stage_state_publish /
// stage_cols_publish).  Roles: wave 0 = P0 (state path, one lane per segment), wave 1 = P1 (derivative part of the record, one stage
// behind P0, one lane per segment), waves 2.. = consumers (one lane per sensitivity column, one stage behind P1).  One barrier per tick;
// stage n of a group lives in ring slot n % 3 (P0 writes it at tick n, P1 completes it at tick n + 1, the consumers read it at tick n + 2),
// the hand-over record in slot n % 2.  Six consumer wavefronts instead of seven: 18 (aero) / 12 (fins) segments per group.
#ifndef SCVX_K1_SPLIT
#define SCVX_K1_SPLIT 1
#endif
// Segment batches per consumer lane.  With the stage split the consumers still wait half of every tick; with NB = 2 each consumer lane
// carries one sensitivity column of TWO segments (two sets of c / accumulator / stage value), the producers' lanes cover twice the
// segments at no cost in time (a lane per segment, and the roles are latency-bound), and a tick serves 36 (aero) / 24 (fins) segments.
// (Without aerodynamics -- exo + fins -- the split block is already balanced, consumers included: one batch.)
#ifndef SCVX_K1_NB
#define SCVX_K1_NB 2
#endif
template <bool AERO> struct K1Split { static constexpr int NB = AERO ? SCVX_K1_NB : 1; };
template <bool AERO, typename R, typename O = R, bool FIN = false>
__global__ __launch_bounds__(64 * PC_WAVES) void linearize_pcp2_kernel(
    DynP<R> p, long nseg, int K, const R* __restrict__ x, const R* __restrict__ u,
    const R* __restrict__ sigma, R dt, int nsub, R* __restrict__ endpoint,
    O* __restrict__ deriv, const int* __restrict__ skip) {
    static_assert(AERO || FIN, "the exo producer is not the bottleneck");
    constexpr int LPS = K1Map<AERO, FIN>::LPS, SPW = K1Map<AERO, FIN>::SPW;
    constexpr int NU = K1Map<AERO, FIN>::NU, NP = K1Map<AERO, FIN>::NP, DSZ = K1Map<AERO, FIN>::DSZ, HV = K1Map<AERO, FIN>::HV;
    typedef typename Vec2<R>::type VEC2;
    typedef typename Vec2<O>::type OVEC2;
    constexpr int NC = PC_WAVES - 2;
    constexpr int NB = K1Split<AERO>::NB;
    constexpr int NS = NC * SPW * NB;          // segments per group
    constexpr int NR = StageRec<AERO, FIN>::N, NH = HandRec<FIN>::N;
    constexpr int RING_D = 3 * NR * NS, HAND_D = 2 * NH * NS, TILE_D = NC * SPW * DSZ;
    __shared__ __attribute__((aligned(16))) R lds[RING_D + HAND_D + TILE_D];
    R* const hand = lds + RING_D;
    R* const tiles = hand + HAND_D;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
#if defined(SCVX_K1_PROF)
    double k1wait_ = 0.0;
    const unsigned long long k1t0_ = __builtin_amdgcn_s_memtime();
#endif
    const R h = dt / R(nsub);
    const R inv_n = R(1.0) / R(nsub);
    const int T = 4 * nsub;                    // stages per segment
    const long ngrp = (nseg + NS - 1) / NS;
    auto advance = [&](long g) {
        while (g < ngrp && block_unchanged(skip, g * NS, NS, nseg, K)) g += gridDim.x;
        return g;
    };
    long grp = advance(blockIdx.x);

    if (wave == 0) {
        // ---------------- P0: the state path, lane = segment ----------------
        const bool live = lane < NS;
        const int l = live ? lane : 0;
        R nx[14], nu6[2 * NU], nsig = R(0.0);
        auto fetch = [&](long g) {
            long seg = g * NS + l;
            if (seg >= nseg) seg = nseg - 1;
            const long b = seg / K;
            const int k = (int)(seg - b * K);
            const R* xk = x + ((size_t)b * (K + 1) + k) * 14;
            const R* uk = u + ((size_t)b * (K + 1) + k) * NU;
#pragma unroll
            for (int i = 0; i < 14; i++) nx[i] = xk[i];
#pragma unroll
            for (int i = 0; i < 2 * NU; i++) nu6[i] = uk[i];
            nsig = sigma[b];
        };
        if (grp < ngrp) fetch(grp);
        while (grp < ngrp) {
            const long nxt = advance(grp + gridDim.x);
            const long seg = grp * NS + l;
            const bool valid = live && seg < nseg;
            const R sig = nsig;
            R xs[14], xa[14], xt[14];
#pragma unroll
            for (int i = 0; i < 14; i++) { xs[i] = nx[i]; xa[i] = xs[i]; xt[i] = xs[i]; }
            R ukv[NU], upv[NU];
#pragma unroll
            for (int j = 0; j < NU; j++) { ukv[j] = nu6[j]; upv[j] = nu6[NU + j]; }
            int slot = 0, hs = 0;
            for (int s = 0; s < nsub; s++) {
#pragma unroll
                for (int stg = 0; stg < 4; stg++) {
                    const R lkp = (R(s) + (stg == 0 ? R(0.0) : (stg == 3 ? R(1.0) : R(0.5)))) * inv_n;
                    const R lkm = R(1.0) - lkp;
                    R uu[NU];
#pragma unroll
                    for (int j = 0; j < NU; j++) uu[j] = fma(ukv[j], lkm, upv[j] * lkp);
                    struct { R g[14]; } st;
                    stage_state_publish<AERO, FIN>(p, xt, uu, st.g, lds + slot * NR * NS + l, NS, live, hand + hs * NH * NS + l, NS);
                    const R wacc = h * ((stg == 0 || stg == 3) ? (R(1.0) / R(6.0)) : (R(1.0) / R(3.0)));
                    const R wnext = h * (stg == 2 ? R(1.0) : R(0.5));
#pragma unroll
                    for (int i = 0; i < 14; i++) {
                        const R dx = sig * st.g[i];
                        xa[i] = fma(wacc, dx, xa[i]);
                        xt[i] = (stg < 3) ? fma(wnext, dx, xs[i]) : xa[i];
                    }
                    slot = slot == 2 ? 0 : slot + 1;
                    hs ^= 1;
                    K1_BAR();   // tick: stage 4 s + stg is handed to P1
                }
#pragma unroll
                for (int i = 0; i < 14; i++) xs[i] = xa[i];
            }
            K1_BAR();   // P1's last stage
            K1_BAR();   // the consumers' last stage
            if (nxt < ngrp) fetch(nxt);
            if (valid) {
                R* ep = endpoint + (size_t)seg * 14;
#pragma unroll
                for (int i = 0; i < 14; i++) ep[i] = xs[i];
            }
#pragma unroll
            for (int bb = 0; bb < NB; bb++) K1_BAR();  // matches the consumers' tile barriers
            grp = nxt;
        }
#if defined(SCVX_K1_PROF)
        if (blockIdx.x == 0 && lane == 0) { g_k1prof[2 * wave] = k1wait_; g_k1prof[2 * wave + 1] = (double)(__builtin_amdgcn_s_memtime() - k1t0_); }
#endif
        return;
    }
    if (wave == 1) {
        // ---------------- P1: the derivative part of the records, one stage behind P0 ----------------
        const bool live = lane < NS;
        const int l = live ? lane : 0;
        while (grp < ngrp) {
            const long nxt = advance(grp + gridDim.x);
            int slot = 0, hs = 0;
            K1_BAR();   // stage 0 has been handed over
            for (int n = 0; n < T; n++) {
                stage_cols_publish<AERO, FIN>(p, hand + hs * NH * NS + l, NS, lds + slot * NR * NS + l, NS, live);
                slot = slot == 2 ? 0 : slot + 1;
                hs ^= 1;
                K1_BAR();
            }
            K1_BAR();   // the consumers' last stage
#pragma unroll
            for (int bb = 0; bb < NB; bb++) K1_BAR();   // tile barriers
            grp = nxt;
        }
#if defined(SCVX_K1_PROF)
        if (blockIdx.x == 0 && lane == 0) { g_k1prof[2 * wave] = k1wait_; g_k1prof[2 * wave + 1] = (double)(__builtin_amdgcn_s_memtime() - k1t0_); }
#endif
        return;
    }

    // ---------------- consumers: lane = (segment, column), two stages behind P0 ----------------
    const int cw = wave - 2;
    const int sl = lane / LPS;
    const int slot_c = lane - sl * LPS;
    const int col = slot_c;
    const bool lane_live = sl < SPW;
    int ls[NB];                                        // local segment index in the group, per batch
#pragma unroll
    for (int bb = 0; bb < NB; bb++) ls[bb] = (bb * NC + cw) * SPW + (lane_live ? sl : 0);
    const bool is_uk = (col >= 14) && (col < 14 + NU);
    const bool is_up = (col >= 14 + NU) && (col < 14 + 2 * NU);
    const int comp = is_uk ? col - 14 : (is_up ? col - 14 - NU : -1);
    const R gsel = (col == NP - 1) ? R(1.0) : R(0.0);
    R ec[NU];
#pragma unroll
    for (int j = 0; j < NU; j++) ec[j] = (comp == j) ? R(1.0) : R(0.0);
    auto sigma_of = [&](long g, int bb) {
        long seg = g * NS + ls[bb];
        if (seg >= nseg) seg = nseg - 1;
        return sigma[seg / K];
    };
    R nsig[NB];
#pragma unroll
    for (int bb = 0; bb < NB; bb++) nsig[bb] = grp < ngrp ? sigma_of(grp, bb) : R(0.0);
    while (grp < ngrp) {
        const long nxt = advance(grp + gridDim.x);
        const long seg_base = grp * NS;
        R sig[NB];
        R c[NB][14], ca[NB][14], ct[NB][14];
#pragma unroll
        for (int bb = 0; bb < NB; bb++) {
            sig[bb] = nsig[bb];
#pragma unroll
            for (int i = 0; i < 14; i++) { c[bb][i] = (col == i) ? R(1.0) : R(0.0); ca[bb][i] = c[bb][i]; ct[bb][i] = c[bb][i]; }
        }
        K1_BAR();  // P0's first stage
        K1_BAR();  // ... completed by P1
        int slot = 0;
        for (int s = 0; s < nsub; s++) {
#pragma unroll
            for (int stg = 0; stg < 4; stg++) {
                const R lkp = (R(s) + (stg == 0 ? R(0.0) : (stg == 3 ? R(1.0) : R(0.5)))) * inv_n;
                const R lkm = R(1.0) - lkp;
                const R wk = is_uk ? lkm : (is_up ? lkp : R(0.0));
                R wc[NU];
#pragma unroll
                for (int j = 0; j < NU; j++) wc[j] = ec[j] * wk;
                const R wacc = h * ((stg == 0 || stg == 3) ? (R(1.0) / R(6.0)) : (R(1.0) / R(3.0)));
                const R wnext = h * (stg == 2 ? R(1.0) : R(0.5));
#pragma unroll
                for (int bb = 0; bb < NB; bb++) {
                    R dc[14];
                    column_deriv_rec_any<AERO, FIN>(p, lds + slot * NR * NS + ls[bb], NS, ct[bb], wc, gsel, sig[bb], dc);
#pragma unroll
                    for (int i = 0; i < 14; i++) {
                        ca[bb][i] = fma(wacc, dc[i], ca[bb][i]);
                        ct[bb][i] = (stg < 3) ? fma(wnext, dc[i], c[bb][i]) : ca[bb][i];
                    }
                }
                slot = slot == 2 ? 0 : slot + 1;
                K1_BAR();
            }
#pragma unroll
            for (int bb = 0; bb < NB; bb++)
#pragma unroll
                for (int i = 0; i < 14; i++) c[bb][i] = ca[bb][i];
        }
#pragma unroll
        for (int bb = 0; bb < NB; bb++) if (nxt < ngrp) nsig[bb] = sigma_of(nxt, bb);
        // ---- epilogue, batch by batch through this wavefront's LDS tile: columns in -> coalesced 16-byte stores out ----
        R* t = tiles + cw * SPW * DSZ;
#pragma unroll
        for (int bb = 0; bb < NB; bb++) {
            if (lane_live) {
#pragma unroll
                for (int i = 0; i < 14; i++) t[sl * DSZ + col * 14 + i] = c[bb][i];
            }
            K1_BAR();
            const long seg0 = seg_base + (long)(bb * NC + cw) * SPW;
            if (seg0 < nseg) {
                const long rem = nseg - seg0;
                const int nvalid = rem < SPW ? (int)rem : SPW;
                const int n2 = nvalid * HV;
                OVEC2* out = reinterpret_cast<OVEC2*>(deriv + (size_t)seg0 * DSZ);
                const VEC2* src = reinterpret_cast<const VEC2*>(t);
#pragma unroll
                for (int r = 0; r < (SPW * HV + 63) / 64; r++) {
                    const int e = lane + 64 * r;
                    if (e < n2) { const VEC2 v = src[e]; OVEC2 o; o.x = O(v.x); o.y = O(v.y); out[e] = o; }
                }
            }
        }
        grp = nxt;
    }
#if defined(SCVX_K1_PROF)
    if (blockIdx.x == 0 && lane == 0) { g_k1prof[2 * wave] = k1wait_; g_k1prof[2 * wave + 1] = (double)(__builtin_amdgcn_s_memtime() - k1t0_); }
#endif
}

template <bool AERO, typename R, bool FIN = false>
__global__ __launch_bounds__(256) void propagate_kernel(DynP<R> p, long nseg, int K, const R* __restrict__ x,
                                                        const R* __restrict__ u,
                                                        const R* __restrict__ sigma, R dt, int nsub,
                                                        R* __restrict__ xnext) {
    const long seg = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= nseg) return;
    const long b = seg / K;
    const int k = (int)(seg - b * K);
    const R* xk = x + ((size_t)b * (K + 1) + k) * 14;
    constexpr int NU = FIN ? 5 : 3;
    const R* uk = u + ((size_t)b * (K + 1) + k) * NU;
    const R sig = sigma[b];
    R xs[14];
#pragma unroll
    for (int i = 0; i < 14; i++) xs[i] = xk[i];
    R ukv[NU], upv[NU];
#pragma unroll
    for (int j = 0; j < NU; j++) { ukv[j] = uk[j]; upv[j] = uk[NU + j]; }
    const R h = dt / R(nsub);
    const R inv_n = R(1.0) / R(nsub);
    for (int s = 0; s < nsub; s++) {
        R xa[14], xt[14];
#pragma unroll
        for (int i = 0; i < 14; i++) {
            xa[i] = xs[i];
            xt[i] = xs[i];
        }
#pragma unroll
        for (int stg = 0; stg < 4; stg++) {
            const R lkp = (R(s) + (stg == 0 ? R(0.0) : (stg == 3 ? R(1.0) : R(0.5)))) * inv_n;
            const R lkm = R(1.0) - lkp;
            R uu[NU];
#pragma unroll
            for (int j = 0; j < NU; j++) uu[j] = fma(ukv[j], lkm, upv[j] * lkp);
            R g[14];
            rhs_only<AERO, FIN>(p, xt, uu, g);
            const R wacc = h * ((stg == 0 || stg == 3) ? (R(1.0) / R(6.0)) : (R(1.0) / R(3.0)));
            const R wnext = h * (stg == 2 ? R(1.0) : R(0.5));
#pragma unroll
            for (int i = 0; i < 14; i++) {
                const R dx = sig * g[i];
                xa[i] = fma(wacc, dx, xa[i]);
                if (stg < 3) xt[i] = fma(wnext, dx, xs[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 14; i++) xs[i] = xa[i];
    }
    R* o = xnext + (size_t)seg * 14;
#pragma unroll
    for (int i = 0; i < 14; i++) o[i] = xs[i];
}

template <typename R>
hipError_t launch_linearize_simple(const scvx_ctx* ctx, int B, int K, const R* x, const R* u, const R* sigma, R dt,
                                   R* endpoint, R* deriv, hipStream_t st, const int* skip = nullptr) {
    const long nseg = (long)B * K;
    if (nseg == 0) return hipSuccess;
    const int spw = ctx->dyn.aero ? K1Map<true>::SPW : K1Map<false>::SPW;
    const long nwave = (nseg + spw - 1) / spw;
    const unsigned grid = (unsigned)((nwave + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    const DynP<R> dp(ctx->dyn);
    if (ctx->dyn.aero)
        hipLaunchKernelGGL((linearize_kernel<true, R>), dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, st, dp, nseg, K, x, u,
                           sigma, dt, ctx->nsub, endpoint, deriv, skip);
    else
        hipLaunchKernelGGL((linearize_kernel<false, R>), dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, st, dp, nseg, K, x, u,
                           sigma, dt, ctx->nsub, endpoint, deriv, skip);
    return hipGetLastError();
}

template <typename R, typename O = R>
hipError_t launch_linearize_t(const scvx_ctx* ctx, int B, int K, const R* x, const R* u, const R* sigma, R dt, R* endpoint,
                              O* deriv, hipStream_t st, const int* skip = nullptr) {
    const long nseg = (long)B * K;
    if (nseg == 0) return hipSuccess;
    const bool fin = ctx->dyn.fin != 0;
    if constexpr (std::is_same<R, O>::value)
        if (ctx->k1_variant == 0 && !fin) return launch_linearize_simple<R>(ctx, B, K, x, u, sigma, dt, endpoint, deriv, st, skip);
    // aero / fin models from 3 substeps up: the producer's stage split over two wavefronts (linearize_pcp2_kernel), six consumer wavefronts
    const bool split = SCVX_K1_SPLIT != 0 && (fin || ctx->dyn.aero) && ctx->k1_sg != 0 && (ctx->k1_persist < 0 ? ctx->nsub >= 3 : ctx->k1_persist != 0);
    const bool exo_nb = !fin && !ctx->dyn.aero && ctx->k1_sg != 0 && (ctx->k1_persist < 0 ? ctx->nsub >= 3 : ctx->k1_persist != 0);   // the exo persistent kernel with SCVX_K1_NB_EXO batches
    const int ns = (PC_WAVES - (split ? 2 : 1)) * (split ? (ctx->dyn.aero ? K1Split<true>::NB : K1Split<false>::NB) : (exo_nb ? SCVX_K1_NB_EXO : 1))
                   * (fin ? K1Map<true, true>::SPW : (ctx->dyn.aero ? K1Map<true>::SPW : K1Map<false>::SPW));
    // stage-granular pipeline by default (faster at every npts measured: 0.70 -> 0.60 ms at npts 1, 3.05 -> 2.96 ms
    // at npts 10, B = 8192, fp64); SCVX_K1_SG=0 selects the substep-granular form
    const bool sg = ctx->k1_sg != 0;
    const long ngrp = (nseg + ns - 1) / ns;
    const long cap = (long)(ctx->num_cus > 0 ? ctx->num_cus : 256) * PC_BLOCKS_PER_CU;   // persistent blocks, one per CU (LDS and VGPRs allow no more)
    // persistent kernel from 3 substeps up (measured, B = 8192 fp64: npts 10 3.21 -> 2.95 ms; npts 1: 0.59 -> 0.67, npts 2:
    // 0.87 -> 0.90 -- a group is latency-bound there, pipeline fill + epilogue); SCVX_K1_PERSIST = 0 / 1 forces
    const bool persist = (sg || fin) && (ctx->k1_persist < 0 ? ctx->nsub >= 3 : ctx->k1_persist != 0);
    const unsigned grid = (unsigned)(!persist || ngrp < cap ? ngrp : cap);
    const dim3 g(grid), blk(64 * PC_WAVES);
    const DynP<R> dp(ctx->dyn);
    if (split) {
        if (fin) {
            if (ctx->dyn.aero) hipLaunchKernelGGL((linearize_pcp2_kernel<true, R, O, true>), g, blk, 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, endpoint, deriv, skip);
            else hipLaunchKernelGGL((linearize_pcp2_kernel<false, R, O, true>), g, blk, 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, endpoint, deriv, skip);
        } else hipLaunchKernelGGL((linearize_pcp2_kernel<true, R, O, false>), g, blk, 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, endpoint, deriv, skip);
    } else if (fin) {   // control_dim = 5: the stage-granular pipeline (persistent from 3 substeps up), exo or aero
        if (persist) {
            if (ctx->dyn.aero) hipLaunchKernelGGL((linearize_pcp_kernel<true, R, O, true>), g, blk, 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, endpoint, deriv, skip);
            else hipLaunchKernelGGL((linearize_pcp_kernel<false, R, O, true>), g, blk, 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, endpoint, deriv, skip);
        } else {
            if (ctx->dyn.aero) hipLaunchKernelGGL((linearize_pc_kernel<true, true, R, O, true>), g, blk, 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, endpoint, deriv, skip);
            else hipLaunchKernelGGL((linearize_pc_kernel<false, true, R, O, true>), g, blk, 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, endpoint, deriv, skip);
        }
    } else if (persist) {
        if (ctx->dyn.aero) hipLaunchKernelGGL((linearize_pcp_kernel<true, R, O>), g, blk, 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, endpoint, deriv, skip);
        else hipLaunchKernelGGL((linearize_pcp_kernel<false, R, O, false, SCVX_K1_NB_EXO>), g, blk, 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, endpoint, deriv, skip);
    } else if (ctx->dyn.aero) {
        if (sg) hipLaunchKernelGGL((linearize_pc_kernel<true, true, R, O>), g, blk, 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, endpoint, deriv, skip);
        else hipLaunchKernelGGL((linearize_pc_kernel<true, false, R, O>), g, blk, 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, endpoint, deriv, skip);
    } else {
        if (sg) hipLaunchKernelGGL((linearize_pc_kernel<false, true, R, O>), g, blk, 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, endpoint, deriv, skip);
        else hipLaunchKernelGGL((linearize_pc_kernel<false, false, R, O>), g, blk, 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, endpoint, deriv, skip);
    }
    return hipGetLastError();
}

hipError_t launch_linearize(const scvx_ctx* ctx, int B, int K, const double* x, const double* u, const double* sigma,
                            double dt, double* endpoint, double* deriv, hipStream_t st, const int* skip) {
    return launch_linearize_t<double>(ctx, B, K, x, u, sigma, dt, endpoint, deriv, st, skip);
}

// double arithmetic, float derivative tiles (scvx_batch_set_linearization_f32): the endpoint stays double
hipError_t launch_linearize_store_f32(const scvx_ctx* ctx, int B, int K, const double* x, const double* u, const double* sigma,
                                      double dt, double* endpoint, float* deriv, hipStream_t st, const int* skip) {
    return launch_linearize_t<double, float>(ctx, B, K, x, u, sigma, dt, endpoint, deriv, st, skip);
}

// fp32 form of K1 (scvx_linearize_f32): the same kernels instantiated in float arithmetic -- half the bytes, twice the
// vector rate of the fp64 instantiation
hipError_t launch_linearize_f32(const scvx_ctx* ctx, int B, int K, const float* x, const float* u, const float* sigma,
                                float dt, float* endpoint, float* deriv, hipStream_t st) {
    // measured at B = 8192 (profiles/r02_k1.md): up to two substeps the column-per-lane form wins in float (0.37 vs 0.43 ms
    // at npts 1: the producer/consumer pipeline pays a barrier per RK stage), beyond that the producer/consumer form does
    if (ctx->nsub <= 2 && ctx->k1_variant != 0 && !ctx->dyn.fin) return launch_linearize_simple<float>(ctx, B, K, x, u, sigma, dt, endpoint, deriv, st);
    return launch_linearize_t<float>(ctx, B, K, x, u, sigma, dt, endpoint, deriv, st);
}

template <typename R>
hipError_t launch_propagate_t(const scvx_ctx* ctx, int B, int K, const R* x, const R* u, const R* sigma, R dt, R* xnext,
                              hipStream_t st) {
    const long nseg = (long)B * K;
    if (nseg == 0) return hipSuccess;
    const unsigned grid = (unsigned)((nseg + 255) / 256);
    const DynP<R> dp(ctx->dyn);
    if (ctx->dyn.fin) {
        if (ctx->dyn.aero) hipLaunchKernelGGL((propagate_kernel<true, R, true>), dim3(grid), dim3(256), 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, xnext);
        else hipLaunchKernelGGL((propagate_kernel<false, R, true>), dim3(grid), dim3(256), 0, st, dp, nseg, K, x, u, sigma, dt, ctx->nsub, xnext);
    } else if (ctx->dyn.aero)
        hipLaunchKernelGGL((propagate_kernel<true, R>), dim3(grid), dim3(256), 0, st, dp, nseg, K, x, u, sigma, dt,
                           ctx->nsub, xnext);
    else
        hipLaunchKernelGGL((propagate_kernel<false, R>), dim3(grid), dim3(256), 0, st, dp, nseg, K, x, u, sigma, dt,
                           ctx->nsub, xnext);
    return hipGetLastError();
}

hipError_t launch_propagate(const scvx_ctx* ctx, int B, int K, const double* x, const double* u, const double* sigma,
                            double dt, double* xnext, hipStream_t st) {
    return launch_propagate_t<double>(ctx, B, K, x, u, sigma, dt, xnext, st);
}
hipError_t launch_propagate_f32(const scvx_ctx* ctx, int B, int K, const float* x, const float* u, const float* sigma,
                                float dt, float* xnext, hipStream_t st) {
    return launch_propagate_t<float>(ctx, B, K, x, u, sigma, dt, xnext, st);
}

#if defined(SCVX_K1_PROF)
extern "C" int scvx_debug_k1_prof(double* out32) {
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_k1prof), sizeof(double) * 32) == hipSuccess ? 0 : -2;
}
#endif

}  // namespace scvx
