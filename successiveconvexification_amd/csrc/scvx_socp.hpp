// The conic subproblem kernel (K4) of libscvx_hip.so: executors, kernel bodies and kernel templates, shared by the two translation
// units that instantiate them -- scvx_batch.hip (control_dim = 3, the reference's model) and scvx_socp_fin.hip (control_dim = 5, the
// fin extension) -- so that the two sets of instantiations compile side by side.
#pragma once
#include <new>
#include "scvx_internal.hpp"
#include "scvx_ipm_core.hpp"

#ifndef SCVX_LDS_FENCE_SCOPE
#define SCVX_LDS_FENCE_SCOPE "wavefront"   // orders the LDS traffic of ONE wavefront (its own tiles): no s_waitcnt needed, the LDS queue of a wavefront is in order.  Cross-wavefront hand-overs go through __syncthreads (BlockEx::sync).  70.9 -> 70.6 ms per launch against "workgroup"
#endif
// WaveExT::chol_inv14 with the inverse riding on the factorisation's column steps (1) instead of following it (0).  MEASURED AND OFF
// (round 5): same results, but chol_inv14 itself 4.2 M -> 5.0 M cycles per solve and the whole factorisation loop 15.8 M -> 20.3 M (28 live
// row registers per lane instead of 14 push the loop's other values out): B = 8192 59.5 -> 63.7 ms per launch, B = 1024 10.15 -> 10.6.
#ifndef SCVX_CHOL_FUSED_INV
#define SCVX_CHOL_FUSED_INV 0
#endif
// WaveExT::chol_inv14's broadcasts: 0 = v_readlane through an SGPR pair (rounds 1-5), 1 = DPP row_newbcast (v_mov_b64_dpp + v_fma_f64),
// 2 = folded into the FMA (v_fmac_f64_dpp, inline asm).  Bit-identical results; measurements in profiles/r06_chol_dpp.md.
#ifndef SCVX_CHOL_DPP
#define SCVX_CHOL_DPP 2
#endif
// order of the trailing updates in the DPP Cholesky's column steps: 0 = all of column J right after its scaling; 1 = the pivot column
// first, the rest in one burst behind the next column's v_rsq_f64; 2 = the rest pinned one by one between the links of the next column's
// pivot chain (fnma_rbc_tied).  Alone on a SIMD: 3,415 / 3,417 / 3,685 ticks per call (profiles/r06_chol_dpp_micro.txt): the order does not matter and pinning costs -- the routine is bound by the issue rate of its 182 64-bit DPP multiply-adds, not by the pivot chains
#ifndef SCVX_CHOL_ORDER
#define SCVX_CHOL_ORDER 1
#endif
#ifndef SCVX_CHAIN_R
#define SCVX_CHAIN_R 8   // steps of the block recurrence whose operands are in flight (10 VGPRs each)
#endif

namespace scvx {

// The tile scratch of socp_kernel (one wavefront per block).  File scope on purpose: every routine of the solver
// reaches it through WaveEx::scratch() as a known LDS symbol, so tile accesses compile to ds_read/ds_write (lgkmcnt
// only).  A pointer carried in the executor object is reloaded from memory in each non-inlined routine, loses its
// address space, and every LDS access becomes a flat_load/flat_store that also waits on the global loads and
// stores in flight (vmcnt) — which serialises the tile arithmetic behind the HBM traffic it is meant to overlap.
// (the tiles of build_kkt(res) -- Sg[42..84), Vn, Pn, Yv: 208 doubles -- exist only when that measured-and-off path is compiled in)
__shared__ __attribute__((aligned(16))) double g_socp_lds[SCVX_FUSED_RES ? 2112 : 1904];   // + the fused border's tiles (Solver::build_kkt: Gn, Rk, Tt, Sg)
// ... and of the fin instantiation (control_dim = 5: 14 x 25 tiles, 24-column [TA | TBm | TBp]); separate symbols so that the
// kernels of the reference's model keep their LDS footprint
__shared__ __attribute__((aligned(16))) double g_socp_lds5[SCVX_FUSED_RES ? 2272 : 2072];
// tile sets of the multi-wavefront factorisations, in doubles (Solver::factor_pipelined needs 3,538 / 3,706, each half of Solver::factor_twisted
// -- since round 6 with the border's node slices, r_k ring, running t and segment scalars -- 3,748 / 3,912: control_dim 3 / 5)
#define SCVX_PIPE_LDS3 3752
#define SCVX_PIPE_LDS5 3912
// ... and the FIRST symbol also holds the two compact tile sets of the two-wavefront two-ended form (Solver::factor_twisted2: 2 x 2,068 / 2 x 2,232)
#define SCVX_PIPE1_LDS3 4160
#define SCVX_PIPE1_LDS5 4480
__shared__ __attribute__((aligned(16))) double g_socp_pipe_lds5[SCVX_PIPE1_LDS5];
__shared__ __attribute__((aligned(16))) double g_socp_pipe_lds25[SCVX_PIPE_LDS5];   // the second tile set (two-ended form: the bottom half)
template <int NU> __device__ __forceinline__ double* socp_lds() { if constexpr (NU == 5) return g_socp_lds5; else return g_socp_lds; }
// The multi-wavefront kernels factorise through the pipeline's own tiles (g_socp_pipe_lds*) and need only the 32-double header of the
// scratch (reduction partials, flags): a symbol of their own, so that they do not carry the single-wavefront kernel's tile space
// (with it the two-wavefront block would not fit four times into a CU's 160 KB)
__shared__ __attribute__((aligned(16))) double g_socp_blk_hdr[32];
// tiles of the two-wavefront factorisation pipeline (multi-wavefront kernels only: a kernel that never references the
// symbol does not get the allocation)
__shared__ __attribute__((aligned(16))) double g_socp_pipe_lds[SCVX_PIPE1_LDS3];   // Solver::factor_pipelined: Sd, So rings | Wb ring (3) | Linv ring (2) | Nf tile | producer tiles
#ifndef SCVX_K4_PIPELINE
#define SCVX_K4_PIPELINE 1
#endif
// second tile set of the TWO-ENDED factorisation (four-wavefront blocks only: wavefronts 2 / 3 eliminate the bottom half of
// the chain upwards while 0 / 1 eliminate the top half downwards); a separate symbol so that the two-wavefront kernel, which
// never references it, keeps its LDS footprint (4 blocks per CU)
__shared__ __attribute__((aligned(16))) double g_socp_pipe_lds2[SCVX_PIPE_LDS3];
#ifndef SCVX_K4_TWISTED
#define SCVX_K4_TWISTED 1
#endif
// the two-ended factorisation for TWO wavefronts per trajectory (Solver::factor_twisted2: each wavefront one half, no hand-over barriers):
// 1 = used by socp_block_kernel<2> instead of the assembly / chain pipeline, 0 = the pipeline (rounds 2-5)
#ifndef SCVX_K4_TWISTED2
#define SCVX_K4_TWISTED2 1
#endif
template <int NU> __device__ __forceinline__ double* socp_pipe_lds() { if constexpr (NU == 5) return g_socp_pipe_lds5; else return g_socp_pipe_lds; }
template <int NU> __device__ __forceinline__ double* socp_pipe_lds2() { if constexpr (NU == 5) return g_socp_pipe_lds25; else return g_socp_pipe_lds2; }

template <int NU_>
struct WaveExT {
    static constexpr int kLanes = 64;
    __device__ __forceinline__ int lane() const { return (int)(threadIdx.x & 63); }   // lane in the wavefront (BlockEx runs tile work on any of its wavefronts)
    __device__ __forceinline__ int nlanes() const { return 64; }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    // LDS-only ordering inside the single wavefront of the block: a release/acquire pair restricted to the local
    // address space compiles to `s_waitcnt lgkmcnt(0)` — global loads (prefetches) and stores stay in flight,
    // whereas __syncthreads() / an unrestricted workgroup fence drains vmcnt(0) at every phase boundary.
    __device__ __forceinline__ void sync_lds() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, SCVX_LDS_FENCE_SCOPE, "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, SCVX_LDS_FENCE_SCOPE, "local");
    }
    __device__ __forceinline__ double sum(double x) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        return x;
    }
    __device__ __forceinline__ double min(double x) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_xor(x, o, 64));
        return x;
    }
    __device__ __forceinline__ bool all(bool b) { return __all(b) != 0; }
    __device__ __forceinline__ double* scratch() { return socp_lds<NU_>(); }

    // value of x in lane `src` (src wave-uniform) delivered to every lane: two v_readlane_b32, no LDS
    static __device__ __forceinline__ double bcast(double x, int src) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
        return __hiloint2double(hi, lo);
    }

    static constexpr int kPrefetchRegs = 1;  // > 0: the next D tile is held in registers (ceil(tile / 64) per lane) while segment k is processed
    static constexpr bool kTwisted = false;
    static constexpr bool kPipelineFactor = false;
    static constexpr bool kFusedResidual = true;   // Solver::build_kkt(res): E'y and E V inside the sequential factorisation loop

    // C(14x14) = (acc ? C : 0) + alpha * A(14 x Kd) B(Kd x 14) on the FP64 matrix pipe: ceil(Kd/4) x
    // v_mfma_f64_16x16x4_f64, tiles in LDS with arbitrary element strides (so transposes are free).
    // Fragment maps (cdna guide §3, f64 form): A: lane l holds A[l&15][l>>4], B: lane l holds B[l>>4][l&15],
    // C/D: register r of lane l is C[(l>>4) + 4r][l&15].  Rows/columns 14,15 and k >= Kd are fed zeros.
    typedef double v4f64 __attribute__((ext_vector_type(4)));
    // nb: number of columns of B / C that exist (right-hand-side blocks of the fused border: 4); the others are fed zeros and not stored
    // Operand fetch of a product with Kd <= 4 KS k-slots: EVERY lane reads from a valid (clamped) LDS address and the value is masked
    // afterwards -- a predicated read compiles to a branch around each ds_read, which puts every read in a basic block of its own and
    // makes the wavefront wait out the LDS latency once per MFMA (round 5: ~850 -> ~350 cycles per 14x14x14 product; the factorisation
    // loop runs nine such products per segment).  All reads of a product are issued before its first MFMA.
    static constexpr int KS = 6;   // k-slots of 4: Kd <= 24 (the widest operand is [TA | TBm | TBp], 24 columns with the fin extension)
    static_assert(14 + 2 * NU_ <= 4 * KS, "fetch_ab / tile_gemm / acc_mac silently truncate a product wider than 4 KS columns");
    // (the Gram products acc_mac(cg, Tc, 1, 4, Tc, 4, 1, 14, ...) read A(row, k) = Tc[row + 4 k] for the clamped rows 0..13 -- up to Tc[65],
    // past the 56-double t slot: harmless because accumulator rows >= 4 are never read, but the LDS behind the slot must hold finite
    // values; every caller zeroes its Gn / Rk / Tt / Sg region in its prologue)
    __device__ __forceinline__ void fetch_ab(const double* A, int sai, int sak, const double* B, int sbk, int sbj, int Kd, int nb,
                                             double (&a)[KS], double (&b)[KS]) {
        const int l = lane();
        const int rc = l & 15, kq = l >> 4;
        const int ra = rc < 14 ? rc : 0, rb = rc < nb ? rc : 0;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            if (4 * s < Kd) {
                const int k = 4 * s + kq, kc = k < Kd ? k : 0;
                a[s] = A[ra * sai + kc * sak];
                b[s] = B[kc * sbk + rb * sbj];
            }
        }
    }
    __device__ __forceinline__ void tile_gemm(double* Cm, int sci, int scj, const double* A, int sai, int sak,
                                              const double* B, int sbk, int sbj, int Kd, double alpha, bool acc, int nb = 14) {
        const int l = lane();
        const int rc = l & 15, kq = l >> 4;
        v4f64 c = {0.0, 0.0, 0.0, 0.0};
        const bool in = rc < 14, inb = rc < nb;
        double a[KS], b[KS];
        fetch_ab(A, sai, sak, B, sbk, sbj, Kd, nb, a, b);
#pragma unroll
        for (int s = 0; s < KS; s++) {
            if (4 * s < Kd) {
                const bool kin = 4 * s + kq < Kd;
                c = __builtin_amdgcn_mfma_f64_16x16x4f64((in && kin) ? a[s] : 0.0, (inb && kin) ? b[s] : 0.0, c, 0, 0, 0);
            }
        }
        if (inb) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = kq + 4 * r;
                if (row < 14) {
                    double* p = Cm + row * sci + rc * scj;
                    *p = (acc ? *p : 0.0) + alpha * c[r];
                }
            }
        }
    }

    // The same product in three parts, for sums of products that share one accumulator (no LDS round trip between the terms):
    //     Acc c; acc_zero(c); acc_mac(c, A1, .., B1, .., K1, alpha1, nb); acc_mac(c, A2, ...); acc_store(c, C, sci, scj, add, nb);
    typedef v4f64 Acc;
    __device__ __forceinline__ void acc_zero(Acc& c) { c = Acc{0.0, 0.0, 0.0, 0.0}; }
    __device__ __forceinline__ void acc_mac(Acc& c, const double* A, int sai, int sak, const double* B, int sbk, int sbj, int Kd,
                                            double alpha, int nb = 14) {
        const int l = lane();
        const int rc = l & 15, kq = l >> 4;
        const bool in = rc < 14, inb = rc < nb;
        double a[KS], b[KS];
        fetch_ab(A, sai, sak, B, sbk, sbj, Kd, nb, a, b);
#pragma unroll
        for (int s = 0; s < KS; s++) {
            if (4 * s < Kd) {
                const bool kin = 4 * s + kq < Kd;
                c = __builtin_amdgcn_mfma_f64_16x16x4f64((in && kin) ? alpha * a[s] : 0.0, (inb && kin) ? b[s] : 0.0, c, 0, 0, 0);
            }
        }
    }
    __device__ __forceinline__ void acc_store(const Acc& c, double* Cm, int sci, int scj, bool add, int nb = 14) {
        const int l = lane();
        const int rc = l & 15, kq = l >> 4;
        if (rc < nb) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = kq + 4 * r;
                if (row < 14) {
                    double* p = Cm + row * sci + rc * scj;
                    *p = (add ? *p : 0.0) + c[r];
                }
            }
        }
    }

    // M = H + diag I + c (14x14, row-major tiles): the pivot tile of the factorisation loop starts from the dense node inverse
    __device__ __forceinline__ void acc_store_init(const Acc& c, double* Cm, const double* H, double diag) {
        const int l = lane();
        const int rc = l & 15, kq = l >> 4;
        if (rc < 14) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = kq + 4 * r;
                if (row < 14) Cm[row * 14 + rc] = (H[row * 14 + rc] + (row == rc ? diag : 0.0)) + c[r];
            }
        }
    }
    // Li = L^-1 (row-major, lower) where L L' = M, for the 14x14 SPD pivot tile in LDS.  Lane i keeps row i of
    // M/L in VGPRs; pivots, column entries and the rows needed by the inversion travel by readlane:
    // 105 broadcasts for the factorisation, 105 for the inverse, no barrier inside.
#if SCVX_CHOL_DPP
    // Round 6: the same factorisation and inverse with every broadcast a DPP row_newbcast inside the 16-lane rows of the wavefront
    // (gfx90a+: `v_mov_b64_dpp ... row_newbcast:n` delivers lane n of each row to the whole row; `v_fmac_f64_dpp` applies it to an
    // operand of the FMA itself).  Lane (l & 15) holds row (l & 15) of M / L, the four rows of 16 lanes work on identical copies.
    // v_readlane went through an SGPR pair: 2 x v_readlane_b32 and the SGPR-read hazard per broadcast, 420 of them per block; here a
    // broadcast is one VALU instruction (SCVX_CHOL_DPP = 1: v_mov_b64_dpp + v_fma_f64) or none at all (= 2: v_fmac_f64_dpp).
    // Same products in the same order as the v_readlane form: bit-identical results.
    template <int N> static __device__ __forceinline__ double rbc(double x) {
        return __builtin_amdgcn_update_dpp(x, x, 0x150 + N, 0xf, 0xf, true);
    }
    // acc -= (value of s in lane N of this row) * b
    template <int N> static __device__ __forceinline__ void fnma_rbc(double& acc, double s, double b) {
#if SCVX_CHOL_DPP == 2
        asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s), "v"(b), "n"(N));
#else
        acc = fma(-rbc<N>(s), b, acc);
#endif
    }
    template <int J, int C> struct CholTrail {   // m[c] -= L[i][J] L[c][J] for c = C .. 13
        static __device__ __forceinline__ void run(double (&m)[14]) {
            if constexpr (C < 14) { fnma_rbc<C>(m[C], m[J], m[J]); CholTrail<J, C + 1>::run(m); }
        }
    };
    // acc -= (value of s in lane N of this row) * b, pinned into a dependent chain: `tie` passes THROUGH the instruction as an in / out
    // operand it does not touch, so the compiler must place it after the instruction that produced `tie` and before the one that
    // consumes it -- the only way to tell the scheduler where an inline-asm instruction goes (it cannot see what it is)
    template <int N> static __device__ __forceinline__ void fnma_rbc_tied(double& acc, double s, double b, double& tie) {
#if SCVX_CHOL_DPP == 2
        asm("v_fmac_f64_dpp %0, -%2, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf" : "+v"(acc), "+v"(tie) : "v"(s), "v"(b), "n"(N));
#else
        acc = fma(-rbc<N>(s), b, acc);
#endif
    }
    template <int J, int C, int STRIDE> struct CholTrailTied {   // m[c] -= L[i][J] L[c][J] for c = C, C + STRIDE, ... <= 13, tied to `tie`
        static __device__ __forceinline__ void run(double (&m)[14], double& tie) {
            if constexpr (C < 14) { fnma_rbc_tied<C>(m[C], m[J], m[J], tie); CholTrailTied<J, C + STRIDE, STRIDE>::run(m, tie); }
        }
    };
    // Column step J, software-pipelined by hand: the trailing update of column J - 1 on the PIVOT column (c = J) was issued at the end of
    // step J - 1; step J runs the pivot chain (row broadcast, v_rsq_f64, two Newton steps, the scaling: nine dependent instructions,
    // ~100 cycles) and the other trailing updates of column J - 1 (c > J) are issued BETWEEN those instructions, each pinned to one link of
    // the chain (fnma_rbc_tied): left to itself the scheduler clusters them ahead of the chain, which then runs with nothing beside it.
    template <int J> struct CholCol {
        static constexpr int P = J > 0 ? J - 1 : 0;   // the column whose trailing updates ride along
        static __device__ __forceinline__ void run(double (&m)[14], double floor_, int i, bool& ok) {
            if constexpr (J < 14) {
                const double d0 = rbc<J>(m[J]);
                ok = ok && (d0 == d0);
                const double d = fmax(d0, floor_);
                double ip = __builtin_amdgcn_rsq(d);
#if SCVX_CHOL_ORDER == 2
                if constexpr (J > 0) CholTrailTied<P, J + 1, 9>::run(m, ip);
#elif SCVX_CHOL_ORDER == 1
                if constexpr (J > 0) CholTrail<P, J + 1>::run(m);   // deferred: columns c > J of the previous step, in one burst behind v_rsq_f64
#endif
                double nhd = -0.5 * d;
#if SCVX_CHOL_ORDER == 2
                if constexpr (J > 0) CholTrailTied<P, J + 2, 9>::run(m, nhd);
#endif
                double t = nhd * ip;
#if SCVX_CHOL_ORDER == 2
                if constexpr (J > 0) CholTrailTied<P, J + 3, 9>::run(m, t);
#endif
                t = fma(t, ip, 0.5);
#if SCVX_CHOL_ORDER == 2
                if constexpr (J > 0) CholTrailTied<P, J + 4, 9>::run(m, t);
#endif
                ip = fma(ip, t, ip);
#if SCVX_CHOL_ORDER == 2
                if constexpr (J > 0) CholTrailTied<P, J + 5, 9>::run(m, ip);
#endif
                t = nhd * ip;
#if SCVX_CHOL_ORDER == 2
                if constexpr (J > 0) CholTrailTied<P, J + 6, 9>::run(m, t);
#endif
                t = fma(t, ip, 0.5);
#if SCVX_CHOL_ORDER == 2
                if constexpr (J > 0) CholTrailTied<P, J + 7, 9>::run(m, t);
#endif
                ip = fma(ip, t, ip);
#if SCVX_CHOL_ORDER == 2
                if constexpr (J > 0) CholTrailTied<P, J + 8, 9>::run(m, ip);
#endif
                double sc = m[J] * ip;
#if SCVX_CHOL_ORDER == 2
                if constexpr (J > 0) CholTrailTied<P, J + 9, 9>::run(m, sc);
#endif
                // The diagonal entry L[J][J] is never read again (the trailing updates and the inverse only use the strictly lower
                // triangle), so lane J keeps 1 / L[J][J] in its place: the inverse fetches it from there by one more row broadcast, and
                // the 14 reciprocals cost neither VGPRs nor SGPRs (the v_readlane form held them in 28 VGPRs).
                m[J] = (i == J) ? ip : sc;
#if SCVX_CHOL_DPP == 2
                // a VALU write of a VGPR must be two wait states ahead of a DPP read of it, and the hazard recogniser does not look inside
                // asm: the column passes THROUGH the nop (in / out operand), so every v_fmac_f64_dpp below depends on it
                asm volatile("s_nop 1" : "+v"(m[J]));
#endif
#if SCVX_CHOL_ORDER == 0
                CholTrail<J, J + 1>::run(m);   // the whole trailing update of this column, at once
#else
                if constexpr (J + 1 < 14) fnma_rbc<J + 1>(m[J + 1], m[J], m[J]);   // the next pivot's column first
#endif
                CholCol<J + 1>::run(m, floor_, i, ok);
            }
        }
    };
    template <int A, int T> struct InvAcc {   // acc -= L[A][t] x[t] for t = T .. A-1
        static __device__ __forceinline__ void run(double& acc, const double (&m)[14], const double (&x)[14]) {
            if constexpr (T < A) { fnma_rbc<A>(acc, m[T], x[T]); InvAcc<A, T + 1>::run(acc, m, x); }
        }
    };
    template <int A> struct InvRow {
        static __device__ __forceinline__ void run(const double (&m)[14], double (&x)[14], int i) {
            if constexpr (A < 14) {
                // (row A's terms t < A - 1 do not depend on x[A - 1]: the scheduler runs them under the previous row's tail)
                double acc = (A == i) ? 1.0 : 0.0;
                InvAcc<A, 0>::run(acc, m, x);
                x[A] = (A >= i) ? acc * rbc<A>(m[A]) : 0.0;
                InvRow<A + 1>::run(m, x, i);
            }
        }
    };
    __device__ __forceinline__ bool chol_inv14(const double* M, double* Li) {
        const int i = lane() & 15;
        const int r = i < 14 ? i : 13;
        double m[14];
#pragma unroll
        for (int c = 0; c < 14; c++) m[c] = M[14 * r + c];
        double dmax = 0.0;
#pragma unroll
        for (int j = 0; j < 14; j++) dmax = fmax(dmax, M[15 * j]);
        const double floor_ = fmax(1e-13 * dmax, 1e-300);
        bool ok = dmax > 0.0;
        CholCol<0>::run(m, floor_, i, ok);
        double x[14];
        InvRow<0>::run(m, x, i);
        if (lane() < 14) {
#pragma unroll
            for (int a = 0; a < 14; a++) Li[14 * a + i] = x[a];
        }
        return ok;
    }
#else
    __device__ __forceinline__ bool chol_inv14(const double* M, double* Li) {
        const int i = lane();
        const int r = i < 14 ? i : 13;
        double m[14];
#pragma unroll
        for (int c = 0; c < 14; c++) m[c] = M[14 * r + c];
        // dynamic regularisation (same rule as the host executor): pivots below 1e-13 * max diagonal are clamped
        double dmax = 0.0;
#pragma unroll
        for (int j = 0; j < 14; j++) dmax = fmax(dmax, M[15 * j]);   // same address in every lane: LDS broadcast reads
        const double floor_ = fmax(1e-13 * dmax, 1e-300);
        bool ok = dmax > 0.0;
#if SCVX_CHOL_FUSED_INV
        // ONE sweep (round 5): lane i also carries row i of X = L^-1 (started as e_i) and every column step j of the factorisation
        // applies its forward-substitution update  X_i -= L[i][j] X_j  (X_j = row j, final at step j, times 1 / L[j][j]) next to the
        // trailing update of M.  Same products in the same order as the two-phase form (factorise, then invert column by column),
        // but the 105 inverse updates no longer wait for the factorisation to end: they fill the issue slots under the pivot chains.
        double x[14];
#pragma unroll
        for (int c = 0; c < 14; c++) x[c] = (c == i) ? 1.0 : 0.0;
        double myip = 0.0;
#pragma unroll
        for (int j = 0; j < 14; j++) {
            const double d0 = bcast(m[j], j);
            ok = ok && (d0 == d0);
            const double d = fmax(d0, floor_);
            double ip = __builtin_amdgcn_rsq(d);
            const double hd = 0.5 * d;
            ip = fma(ip, fma(-hd * ip, ip, 0.5), ip);
            ip = fma(ip, fma(-hd * ip, ip, 0.5), ip);
            if (i == j) myip = ip;
            m[j] = (i == j) ? d * ip : m[j] * ip;       // L[i][j]; rows above j hold junk in column j, never read
            const double l0 = (i > j) ? m[j] : 0.0;     // rows <= j of X are final: they take no update
#pragma unroll
            for (int c = 0; c <= j; c++) {
                const double xjc = bcast(x[c], j) * ip;   // X[j][c]
                x[c] = fma(-l0, xjc, x[c]);
            }
#pragma unroll
            for (int c = j + 1; c < 14; c++) {
                const double lcj = bcast(m[j], c);
                m[c] = fma(-m[j], lcj, m[c]);   // rows i < c update junk (their upper triangle is never read): no select
            }
        }
        if (i < 14) {
#pragma unroll
            for (int c = 0; c < 14; c++) Li[14 * i + c] = (c <= i) ? x[c] * myip : 0.0;
        }
        return ok;
#else
        double ipv[14];   // 1 / L[j][j], identical in every lane: the inverse below multiplies instead of dividing
#pragma unroll
        for (int j = 0; j < 14; j++) {
            const double d0 = bcast(m[j], j);
            ok = ok && (d0 == d0);
            const double d = fmax(d0, floor_);          // NaN-safe: fmax returns the non-NaN operand
            // 1/sqrt(d) from v_rsq_f64 and two Newton steps (d is a clamped positive pivot: no special cases) — the
            // library sqrt + division pair is ~50 dependent instructions on this critical path, this is 9
            double ip = __builtin_amdgcn_rsq(d);
            const double hd = 0.5 * d;
            ip = fma(ip, fma(-hd * ip, ip, 0.5), ip);
            ip = fma(ip, fma(-hd * ip, ip, 0.5), ip);
            ipv[j] = ip;
            m[j] = (i == j) ? d * ip : m[j] * ip;       // rows above j hold junk in column j, never read
#pragma unroll
            for (int c = j + 1; c < 14; c++) {
                const double lcj = bcast(m[j], c);
                m[c] = fma(-m[j], lcj, m[c]);   // rows i < c update junk (their upper triangle is never read): no select
            }
        }
        // inverse: lane c owns column c of L^-1: x[i] = (delta_ic - sum_{t=c}^{i-1} L[i][t] x[t]) / L[i][i]
        double x[14];
#pragma unroll
        for (int a = 0; a < 14; a++) {
            double acc = (a == i) ? 1.0 : 0.0;
#pragma unroll
            for (int t = 0; t < a; t++) {
                const double lat = bcast(m[t], a);  // L[a][t]
                acc = fma(-lat, x[t], acc);         // x[t] = 0 for t < i: no select
            }
            x[a] = (a >= i) ? acc * ipv[a] : 0.0;
        }
        if (i < 14) {
#pragma unroll
            for (int a = 0; a < 14; a++) Li[14 * a + i] = x[a];
        }
        return ok;
#endif
    }
#endif   // SCVX_CHOL_DPP

    // out_k = z_k + N_k out_{k-1} (forward) / out_k = z_k + N_{k+1}' out_{k+1} (reverse) for NR right-hand sides at
    // once: the only sequential part of the block-tridiagonal solve, run entirely on the FP64 matrix pipe.
    // (N_k is the NEGATED coupling tile, stored TRANSPOSED: element (i, j) at 14 j + i — see Solver::S_solve.)
    //
    // One step is  D = Z_k E + N_k T_{k-1}  as five v_mfma_f64_16x16x4:  T holds the NR running 14-vectors as columns
    // 0..NR-1.  With the fragment maps of tile_gemm, register r of lane (g = l>>4, n = l&15) of the result is
    // D[g + 4r][n]; feeding MFMA c the k-slots {g + 4c} makes its B operand B[g + 4c][n] = register c of the SAME lane
    // of the previous result — the recurrence never leaves the accumulator registers: no LDS, no cross-lane traffic,
    // no barrier.  The A operands are plain loads from the tile (forward, lane (g, row): element (g + 4c) * 14 + row,
    // 4 runs of 14 consecutive doubles per instruction; reverse: the transposed element 14 row + g + 4c), and z rides in a fifth MFMA against a constant selector
    // (A[row][16 + g] = z_g[row], B[16 + g][n] = delta(g, n)), issued ahead of the dependent four.  Operands for step
    // s + R are requested while step s runs (R x 10 VGPRs in flight).
    template <int NR, class NP>
    __device__ __forceinline__ void chain_n(int K, const ipm::cgptr (&z)[NR], NP N, const ipm::gptr (&o)[NR],
                                               bool reverse) {
        chain_range_n<NR>(K, z, N, o, reverse, reverse ? K - 1 : 0, K, true);
    }
    // The same recurrence over ns nodes starting at node k0 (the first one without a coupling term), for the two-ended
    // solve of BlockEx<4>: K is only the number of tiles (the reverse form reads the tile of node k + 1).
    // NP: pointer to the coupling tiles in the factor's storage type (double, or float: Solver's FStor); operands widen on load
    template <int NR, class NP>
    __device__ __forceinline__ void chain_range_n(int K, const ipm::cgptr (&z)[NR], NP N, const ipm::gptr (&o)[NR],
                                                     bool reverse, int k0, int ns, bool store_first) {
        static_assert(NR >= 1 && NR <= 4, "right-hand sides ride in k-slots 16..19");
        constexpr int R = SCVX_CHAIN_R;
        const int l = lane(), n = l & 15, g = l >> 4;
        const bool rin = n < 14;   // for the A operands n is the tile row
        // every lane loads from a valid (clamped) address and masks the value afterwards: a predicated load would
        // compile to a branch around each of the five loads of a step
        int offA[4];
        double mA[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int kk = g + 4 * c;
            const bool in = rin && kk < 14;
            offA[c] = in ? (reverse ? 14 * n + kk : kk * 14 + n) : 0;
            mA[c] = in ? 1.0 : 0.0;
        }
        const bool zin = rin && g < NR;
        const int offZ = zin ? n : 0;
        ipm::cgptr zp = z[0];
        ipm::gptr op = o[0];
#pragma unroll
        for (int q = 1; q < NR; q++) {
            if (g == q) zp = z[q];
            if (n == q) op = o[q];
        }
        const double bsel = (g == n) ? 1.0 : 0.0;
        const bool oin = n < NR;
        const bool oin3 = oin && g < 2;   // register 3 holds row g + 12
        const int dk = reverse ? -1 : 1;
        // the tile operands stay in the factor's storage type until the step that consumes them: a conversion at the load would make
        // every request wait for its own data and undo the R-deep prefetch
        typedef decltype(+(*N)) NT;
        NT st[R][4];
        double sz[R];
        auto issue = [&](int s, NT (&f)[4], double& fz) {
            const int k = k0 + dk * s;
            const NP base = N + (size_t)(reverse ? (k + 1 < K ? k + 1 : k) : k) * 196;   // reverse: the tile of node k + 1
#pragma unroll
            for (int c = 0; c < 4; c++) f[c] = base[offA[c]];
            fz = zp[14 * k + offZ];
        };
#pragma unroll
        for (int q = 0; q < R; q++)
            if (q < ns) issue(q, st[q], sz[q]);
        v4f64 d = {0.0, 0.0, 0.0, 0.0};
        for (int s0 = 0; s0 < ns; s0 += R) {
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int s = s0 + q;
                if (s < ns) {
                    // the first node has no coupling term (its tile is never written: mask, don't multiply)
                    const double m0 = s > 0 ? 1.0 : 0.0;
                    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(zin ? sz[q] : 0.0, bsel, acc, 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const double a = (mA[c] * m0 != 0.0) ? (double)st[q][c] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, d[c], acc, 0, 0, 0);
                    }
                    d = acc;
                    // operands of step s + R go into the registers this step has just consumed
                    if (s + R < ns) issue(s + R, st[q], sz[q]);
                    const int k = k0 + dk * s;
                    const bool st_ = store_first || s > 0;
                    if (oin && st_) {
                        op[14 * k + g] = d[0];
                        op[14 * k + g + 4] = d[1];
                        op[14 * k + g + 8] = d[2];
                    }
                    if (oin3 && st_) op[14 * k + g + 12] = d[3];
                }
            }
        }
    }
    template <class NP>
    __device__ __forceinline__ void chain(int K, ipm::cgptr z, NP N, ipm::gptr out, bool reverse) {
        const ipm::cgptr zs[1] = {z};
        const ipm::gptr os[1] = {out};
        chain_n<1>(K, zs, N, os, reverse);
    }
};

// Small batches: NW wavefronts cooperate on ONE trajectory (a lone wavefront per trajectory leaves most of the chip idle
// below a few hundred trajectories and each solve is latency-bound: 26 ms per subproblem).  The streaming sweeps, the
// E / E' products and the node loops of the solver are written against lane()/nlanes(), so they simply spread over
// 64 NW lanes; the 14x14 tile arithmetic and the block recurrences stay on wavefront 0 (they are sequential in k), with
// workgroup barriers where the single-wavefront executor needs none.  Reductions go through LDS in a fixed order, so
// every lane of the workgroup sees bit-identical scalars (uniform control flow, as in WaveEx).
typedef WaveExT<3> WaveEx;

template <int NW, int NU_ = 3>
struct BlockEx {
    WaveExT<NU_> w0;
    static constexpr int kLanes = 64 * NW;
    static constexpr int kPrefetchRegs = 1;
    static constexpr bool kFusedResidual = false;  // the multi-wavefront factorisation pipelines keep the separate residual passes
    // the factorisation loop as a producer / consumer pair of wavefronts (Solver::factor_pipelined)
    static constexpr bool kPipelineFactor = SCVX_K4_PIPELINE != 0;
    // two-ended (twisted) factorisation and solve: the chain is eliminated from both ends towards the middle block by two
    // producer / consumer pairs, and the solve's recurrences run on wavefronts 0 and 2 side by side (Solver::factor_twisted)
    static constexpr bool kTwisted = SCVX_K4_PIPELINE != 0 && ((NW == 4 && SCVX_K4_TWISTED != 0) || (NW == 2 && SCVX_K4_TWISTED2 != 0));
    static constexpr int kPipeDoubles = NU_ == 5 ? SCVX_PIPE_LDS5 : SCVX_PIPE_LDS3;   // capacity of each of the two tile sets
    static constexpr int kPipe1Doubles = NU_ == 5 ? SCVX_PIPE1_LDS5 : SCVX_PIPE1_LDS3;   // capacity of the first symbol
    __device__ __forceinline__ double* pipe_scratch2() { return socp_pipe_lds2<NU_>(); }
    template <int NR, class NP>
    __device__ __forceinline__ void chain_range_n(int wv, int K, const ipm::cgptr (&z)[NR], NP N, const ipm::gptr (&o)[NR],
                                                     bool reverse, int k0, int ns, bool store_first) {
        // wv: 0 = the top half's recurrences, 2 = the bottom half's (wavefront 2 of a four-wavefront block, wavefront 1 of a two-wavefront one)
        if (wave() == (NW == 2 ? wv >> 1 : wv)) w0.template chain_range_n<NR>(K, z, N, o, reverse, k0, ns, store_first);
    }
    __device__ __forceinline__ int wave() const { return (int)(threadIdx.x >> 6); }
    __device__ __forceinline__ int wlane() const { return (int)(threadIdx.x & 63); }
    __device__ __forceinline__ double* pipe_scratch() { return socp_pipe_lds<NU_>(); }
    __device__ __forceinline__ void w_sync_lds() { w0.sync_lds(); }
    __device__ __forceinline__ void w_tile_gemm(double* Cm, int sci, int scj, const double* A, int sai, int sak, const double* B,
                                                int sbk, int sbj, int Kd, double alpha, bool acc) {
        w0.tile_gemm(Cm, sci, scj, A, sai, sak, B, sbk, sbj, Kd, alpha, acc);
    }
    __device__ __forceinline__ bool w_chol_inv14(const double* M, double* Li) { return w0.chol_inv14(M, Li); }
    // wave-level accumulator products (the calling wavefront, whichever it is: the stages of the factorisation pipeline)
    typedef typename WaveExT<NU_>::Acc WAcc;
    __device__ __forceinline__ void w_acc_zero(WAcc& c) { w0.acc_zero(c); }
    __device__ __forceinline__ void w_acc_mac(WAcc& c, const double* A, int sai, int sak, const double* B, int sbk, int sbj, int Kd,
                                              double alpha, int nb = 14) { w0.acc_mac(c, A, sai, sak, B, sbk, sbj, Kd, alpha, nb); }
    __device__ __forceinline__ void w_acc_store(const WAcc& c, double* Cm, int sci, int scj, bool add, int nb = 14) { w0.acc_store(c, Cm, sci, scj, add, nb); }
    __device__ __forceinline__ void w_acc_store_init(const WAcc& c, double* Cm, const double* H, double diag) { w0.acc_store_init(c, Cm, H, diag); }
    __device__ __forceinline__ void acc_store_init(const WAcc& c, double* Cm, const double* H, double diag) { if (first()) w0.acc_store_init(c, Cm, H, diag); }
    __device__ __forceinline__ int lane() const { return (int)threadIdx.x; }
    __device__ __forceinline__ int nlanes() const { return 64 * NW; }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ void sync_lds() { __syncthreads(); }
    __device__ __forceinline__ double* hdr() { if constexpr (kPipelineFactor) return g_socp_blk_hdr; else return socp_lds<NU_>(); }
    __device__ __forceinline__ double* scratch() { return hdr(); }   // the tiles behind the header exist only without the pipeline (sequential build_kkt)
    __device__ __forceinline__ bool first() const { return threadIdx.x < 64; }
    // slots 0..NW-1 of the scratch header hold the per-wavefront partials, slot 16 a flag
    template <class OP>
    __device__ __forceinline__ double reduce(double x, OP op) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x = op(x, __shfl_xor(x, o, 64));
        double* red = hdr();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
        __syncthreads();
        double r = red[0];
#pragma unroll
        for (int q = 1; q < NW; q++) r = op(r, red[q]);
        __syncthreads();   // the partials may be overwritten by the next reduction
        return r;
    }
    __device__ __forceinline__ double sum(double x) { return reduce(x, [](double a, double b) { return a + b; }); }
    __device__ __forceinline__ double min(double x) { return reduce(x, [](double a, double b) { return fmin(a, b); }); }
    __device__ __forceinline__ bool all(bool b) { return __syncthreads_and(b ? 1 : 0) != 0; }
    __device__ __forceinline__ void tile_gemm(double* Cm, int sci, int scj, const double* A, int sai, int sak, const double* B,
                                              int sbk, int sbj, int Kd, double alpha, bool acc, int nb = 14) {
        if (first()) w0.tile_gemm(Cm, sci, scj, A, sai, sak, B, sbk, sbj, Kd, alpha, acc, nb);
    }
    typedef typename WaveExT<NU_>::Acc Acc;
    __device__ __forceinline__ void acc_zero(Acc& c) { w0.acc_zero(c); }
    __device__ __forceinline__ void acc_mac(Acc& c, const double* A, int sai, int sak, const double* B, int sbk, int sbj, int Kd,
                                            double alpha, int nb = 14) {
        if (first()) w0.acc_mac(c, A, sai, sak, B, sbk, sbj, Kd, alpha, nb);
    }
    __device__ __forceinline__ void acc_store(const Acc& c, double* Cm, int sci, int scj, bool add, int nb = 14) {
        if (first()) w0.acc_store(c, Cm, sci, scj, add, nb);
    }
    __device__ __forceinline__ bool chol_inv14(const double* M, double* Li) {
        double* flag = hdr() + 16;
        if (first()) {
            const bool ok = w0.chol_inv14(M, Li);
            if (threadIdx.x == 0) *flag = ok ? 1.0 : 0.0;
        }
        __syncthreads();
        const bool ok = *flag != 0.0;
        __syncthreads();
        return ok;
    }
    template <class NP>
    __device__ __forceinline__ void chain(int K, ipm::cgptr z, NP N, ipm::gptr out, bool reverse) {
        if (first()) w0.chain(K, z, N, out, reverse);
    }
    template <int NR, class NP>
    __device__ __forceinline__ void chain_n(int K, const ipm::cgptr (&z)[NR], NP N, const ipm::gptr (&o)[NR], bool reverse) {
        if (first()) w0.template chain_n<NR>(K, z, N, o, reverse);
    }
};

// Running totals over the solve_steps enqueued since the last scvx_batch_get_step_stats (one atomic per trajectory per
// step): what a timed region actually executed -- conic solves, their interior-point iterations, how many were warm-started or
// skipped, how many steps were rejected / failed.
enum { ACC_TRAJ_STEPS = 0, ACC_SOLVES, ACC_IPM_ITERS, ACC_WARM, ACC_SKIPPED, ACC_REJECTED, ACC_FAILED, ACC_CONVERGED, ACC_N };

// info[b] = {status, iters, merit, pobj}
// DS: element type of the linearisation the discretisation kernel wrote (double; float behind scvx_batch_set_linearization_f32)
template <class Ex, class DS = double, int NU = 3>
__device__ __forceinline__ void socp_body(const ipm::Consts& Cin, int B, size_t work_stride, const double* x, const double* u,
                                          const double* endpoint, const DS* deriv, const double* rk, const double* ic,
                                          const int* active, double* work, double* sol, double* nu, double* info,
                                          const int* step_status, double* ttr, double* acc) {
    const int b = blockIdx.x;
    if (b >= B) return;
    if (active && !active[b]) return;
    // Opt-in shortcut (scvx_solver_opts.reuse_inactive_tr): after a REJECTED step the subproblem is the same one with a
    // halved radius (rocketland.jl:299-301 keeps about / dynam).  If the optimum just found lies strictly inside the new
    // radius, the radius row is inactive with a zero multiplier and that optimum still satisfies every KKT condition of
    // the new subproblem: the solve would return it again.  sol / nu / info are left as they are; iters = 0 marks it.
    if (Cin.pad && step_status[b] == SCVX_ST_REJECTED && ttr[b] <= (1.0 - 1e-6) * rk[b]) {
        if (threadIdx.x == 0) { info[4 * b + 1] = 0.0; atomicAdd(acc + ACC_SKIPPED, 1.0); }
        return;
    }
#if defined(SCVX_K4_STAGGER_US)
    // experiment (round 5, +-0.5 %: profiles/r05_k4_byte_budget.md section 4, last paragraph): the 2,048 wavefronts of a launch's first round start together and run the same phases at the
    // same time; a start delay spread over one interior-point iteration de-correlates them
    if (blockIdx.x < 2048) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)(((blockIdx.x * 2654435761u) >> 16) % 16u) * (unsigned long long)(SCVX_K4_STAGGER_US * 100 / 16);
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(64);
    }
#endif
    const int K = Cin.K;
    // The solver object (some 60 pointers into the slab, the layout, the scalars of the current factorisation), the executor and
    // the constants live in LDS, one copy per wavefront.  As an automatic object it sat in private memory -- 1.4 KB per LANE,
    // 64 identical copies per wavefront, 0.2 GB for the trajectories in flight -- and every non-inlined routine of the solver
    // opens with loads of the members it needs: some 750 of them per interior-point iteration, each a 256 / 512-byte row of that
    // private memory that has long left L2 when the routine comes round again.  Every lane writes the same values.
    typedef ipm::Solver<Ex, double, DS, NU, SCVX_FACTOR_T> SolverT;
    struct Frame {
        ipm::Consts C; Ex ex; SolverT S;
        __device__ Frame(const ipm::Consts& c) : C(c), ex(), S(ex, C) {}
    };
    constexpr int NWV = Ex::kLanes / 64;
    constexpr int FSZ = (int)((sizeof(Frame) + 15) & ~(size_t)15);
    __shared__ __attribute__((aligned(16))) unsigned char frame_mem[NWV * FSZ];
    Frame* const F = new (frame_mem + (threadIdx.x >> 6) * FSZ) Frame(Cin);
    const ipm::Consts& C = F->C;
    Ex& ex = F->ex;
    SolverT& S = F->S;
    // kernel arguments are HBM pointers: hand them to the solver typed as such (see ipm::gptr)
    // warm start: the last solve in this slab was for the same about / dynam (its step was rejected) and is still valid
    const bool warm = C.warm && step_status[b] == SCVX_ST_REJECTED && ttr[b] < 1e300;
    const ipm::Result r = S.solve((ipm::cgptr)(x + (size_t)b * (K + 1) * 14), (ipm::cgptr)(u + (size_t)b * (K + 1) * NU),
                                  (ipm::cgptr)(endpoint + (size_t)b * K * 14), (typename ipm::gp<DS>::cptr)(deriv + (size_t)b * K * (14 * (14 + 2 * NU + 1))), rk[b],
                                  (ipm::cgptr)(ic + (size_t)b * 6), (ipm::gptr)(work + (size_t)b * work_stride), warm);
    const int nxu = S.L.nx + S.L.nu_;
    const int nl = ex.nlanes();
    double* so = sol + (size_t)b * (nxu + 1);
    for (int i = threadIdx.x; i < nxu; i += nl) so[i] = S.V[i];
    double* no = nu + (size_t)b * S.L.ny;
    for (int i = threadIdx.x; i < S.L.ny; i += nl) no[i] = S.V[nxu + i];
    if (threadIdx.x == 0) {
        so[nxu] = S.V[S.L.iS];
        info[4 * b + 0] = (double)r.status;
        info[4 * b + 1] = (double)r.iters;
        info[4 * b + 2] = r.merit;
        info[4 * b + 3] = r.pobj;
        ttr[b] = S.V[S.L.iTTR];   // the trust-region norm bound at the optimum (Jtr of build_model)
        atomicAdd(acc + ACC_SOLVES, 1.0);
        atomicAdd(acc + ACC_IPM_ITERS, (double)r.iters);
        if (r.warmed) atomicAdd(acc + ACC_WARM, 1.0);
#if defined(SCVX_IPM_PROF)
        if (b == 0) for (int i = 0; i < 32; i++) work[i] = S.prof[i];  // diagnostic build: section cycles of trajectory 0
#endif
    }
#if defined(SCVX_IPM_PROF)
    if (b == 0 && threadIdx.x > 0 && (threadIdx.x & 63) == 0 && threadIdx.x < 256)
        for (int i = 0; i < 32; i++) work[32 * (threadIdx.x >> 6) + i] = S.prof[i];  // ... and of its other wavefronts
#endif
}

// one wavefront per trajectory (large batches: the chip is filled by trajectories)
#ifndef SCVX_K4_OCC
// Wavefronts per SIMD the single-wavefront solver is compiled for.  Measured (profiles/r02_k4_sections.md): 2 and 3 give
// the same throughput at B = 8192 (143 vs 142 ms per solve of the batch: the kernel runs at the HBM streaming rate either
// way), 2 is 4 % faster below 3,072 trajectories, and at 2 (<= 256 VGPRs) nothing spills -- at 3 (168 VGPRs) 147
// registers did.  4 (128 VGPRs) is 17 % slower, 1 is 40 % slower at the full batch.
#define SCVX_K4_OCC 2
#endif
template <class DS, int NU>
__global__ __launch_bounds__(64, SCVX_K4_OCC) void socp_kernel_t(ipm::Consts C, int B, size_t work_stride,
                                                  const double* __restrict__ x, const double* __restrict__ u,
                                                  const double* __restrict__ endpoint, const DS* __restrict__ deriv,
                                                  const double* __restrict__ rk, const double* __restrict__ ic,
                                                  const int* __restrict__ active, double* __restrict__ work,
                                                  double* __restrict__ sol, double* __restrict__ nu,
                                                  double* __restrict__ info, const int* __restrict__ step_status,
                                                  double* __restrict__ ttr, double* __restrict__ acc) {
    socp_body<WaveExT<NU>, DS, NU>(C, B, work_stride, x, u, endpoint, deriv, rk, ic, active, work, sol, nu, info, step_status, ttr, acc);
}
// NW wavefronts per trajectory (batches that cannot fill the chip with one wavefront each)
// Compiled for 2 wavefronts per SIMD like socp_kernel (248 VGPRs, no spills; unconstrained the compiler takes 274 = one
// per SIMD, and a batch of more than 1,024 / NW trajectories runs in two rounds: B = 1,024 with NW = 2 took 14.5 ms, now 9.9).
#ifndef SCVX_K4_BLOCK_OCC
#define SCVX_K4_BLOCK_OCC 2
#endif
template <int NW, class DS = double, int NU = 3>
__global__ __launch_bounds__(64 * NW, SCVX_K4_BLOCK_OCC) void socp_block_kernel(ipm::Consts C, int B, size_t work_stride,
                                                  const double* __restrict__ x, const double* __restrict__ u,
                                                  const double* __restrict__ endpoint, const DS* __restrict__ deriv,
                                                  const double* __restrict__ rk, const double* __restrict__ ic,
                                                  const int* __restrict__ active, double* __restrict__ work,
                                                  double* __restrict__ sol, double* __restrict__ nu,
                                                  double* __restrict__ info, const int* __restrict__ step_status,
                                                  double* __restrict__ ttr, double* __restrict__ acc) {
    socp_body<BlockEx<NW, NU>, DS, NU>(C, B, work_stride, x, u, endpoint, deriv, rk, ic, active, work, sol, nu, info, step_status, ttr, acc);
}

// what a launch of the conic solve needs from the batch (scvx_batch.hip fills it; scvx_socp_fin.hip launches the NU = 5 kernels)
struct SocpLaunch {
    ipm::Consts C;
    int B;
    size_t work_stride;
    const double *x, *u, *endpoint, *deriv;
    const float* deriv_f;      // non-null: the derivative tiles are float (scvx_batch_set_linearization_f32)
    const double *rk, *ic;
    const int* mask;
    double *work, *sol, *nu, *info;
    const int* status;
    double *ttr, *acc;
    hipStream_t stream;
};
void launch_socp_fin(const SocpLaunch& a, int waves);   // scvx_socp_fin.hip

}  // namespace scvx
