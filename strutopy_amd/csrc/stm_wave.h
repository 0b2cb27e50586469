// stm_wave.h -- wave64 helpers and scalar semantics shared by the E-step kernels (gfx950).
//
// One wavefront (64 lanes) owns one document.  Vectors of length n = K-1 live one
// component per lane (VPL components per lane when n > 64); scalars of the line search
// are wave-uniform (struct ud: LDS slots), so the solver's control flow compiles to scalar branches.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stm {

constexpr int PROF_SLOTS = 48;   // shader-clock counters per document of the debug profile (STM_DEBUG_PROF; stm_debug_get_prof)


constexpr int WAVE = 64;

__device__ __forceinline__ double uni(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// A wave-uniform double of the solver's scalar state machine, kept in an LDS slot: a state loads what it reads (one
// broadcast ds_read_b64 each, in flight together), computes on VGPR copies and stores what it changes.  fp64 arithmetic is
// VALU-only: pinned to SGPRs instead (rounds 1-2) the ~45 scalars overflowed the SGPR file and every use paid v_readlane /
// v_mov / s_nop plumbing (solver 6.16 -> 5.59 ms over the driver's first twelve iterations); as plain VGPR values they
// would cost 90 registers next to beta_d's 100.
struct ud {
    double *p;
    __device__ __forceinline__ explicit ud(double *slot) : p(slot) {}
    __device__ __forceinline__ ud &operator=(double x) { *p = x; return *this; }
    __device__ __forceinline__ ud &operator=(const ud &o) { *p = *o.p; return *this; }
    __device__ __forceinline__ operator double() const { return *p; }
};
#define STM_UD(ss, n) ud n((ss) + U_##n)

// Uniform read of kernel-lifetime-constant global data through the scalar cache (s_load): the compiler only does this on
// its own when it can prove that nobody writes the buffer, which it cannot for plain pointers in a parameter struct.
template <typename T>
__device__ __forceinline__ T scalar_load(const T *p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}

// value held by lane `src` (uniform src) -> uniform
__device__ __forceinline__ double lane_bcast(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

// ---- wave64 all-reduce on the DPP crossbar (no LDS traffic): 4 intra-row steps
// (quad_perm xor1, xor2, row_half_mirror, row_mirror) leave every lane of a 16-lane row
// holding the row total; the four row totals are then read with v_readlane and combined
// on uniform operands, so the result is wave-uniform (SGPR-resident) by construction.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    // a full permutation inside the rows (all rows and banks enabled): no lane keeps its old destination, so none is
    // set up (with old = the source the compiler copies both halves first)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// row_bcast:15 / row_bcast:31 into the rows of ROWS; the other rows receive FILL
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_bcast(double v, double fill) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, CTRL, ROWS, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, CTRL, ROWS, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // row_half_mirror
constexpr int DPP_MIRROR = 0x140;      // row_mirror
constexpr int DPP_BCAST15 = 0x142;     // row_bcast:15: lane 15 of the row before, to every lane of rows 1 and 3 (row_mask 0xa)
constexpr int DPP_BCAST31 = 0x143;     // row_bcast:31: lane 31, to every lane of rows 2 and 3 (row_mask 0xc)

// wave64 all-reduce: 4 butterfly steps leave every lane of a 16-lane row holding the row total r0..r3; two broadcast steps
// then build (r3 + r2) + (r1 + r0) in row 3 -- bit for bit the (r0 + r1) + (r2 + r3) of four v_readlane and three adds,
// at half the instructions -- and lane 63 hands it out, wave-uniform (SGPR-resident) by construction.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<DPP_XOR1>(v);
    v += dpp_move<DPP_XOR2>(v);
    v += dpp_move<DPP_HALF_MIRROR>(v);
    v += dpp_move<DPP_MIRROR>(v);
    v += dpp_bcast<DPP_BCAST15, 0xa>(v, 0.0);
    v += dpp_bcast<DPP_BCAST31, 0xc>(v, 0.0);
    return lane_bcast(v, 63);
}
// N sums at once: wave_sum()'s steps on every value, step by step (each result has wave_sum()'s bits).  One reduction is six
// dependent move-and-add steps with the DPP's wait states in between; N of them fill each other's gaps instead of queueing up.
template <int N>
__device__ __forceinline__ void wave_sum_n(double (&v)[N]) {
#define STM_WS_STEP(expr)                                                 \
    {                                                                     \
        _Pragma("unroll") for (int q = 0; q < N; ++q) { const double t = (expr); v[q] += t; } \
        _Pragma("unroll") for (int q = 0; q < N; ++q) asm volatile("" : "+v"(v[q]));          \
    }
    STM_WS_STEP(dpp_move<DPP_XOR1>(v[q]))
    STM_WS_STEP(dpp_move<DPP_XOR2>(v[q]))
    STM_WS_STEP(dpp_move<DPP_HALF_MIRROR>(v[q]))
    STM_WS_STEP(dpp_move<DPP_MIRROR>(v[q]))
    STM_WS_STEP((dpp_bcast<DPP_BCAST15, 0xa>(v[q], 0.0)))
    STM_WS_STEP((dpp_bcast<DPP_BCAST31, 0xc>(v[q], 0.0)))
#undef STM_WS_STEP
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = lane_bcast(v[q], 63);
}
// np.max semantics: NaN propagates
__device__ __forceinline__ double nanmax(double a, double b) {
    return __builtin_isunordered(a, b) ? __builtin_nan("") : fmax(a, b);
}
// np.max over the wave (NaN propagates): v_max_f64 ignores NaNs, so reduce with it and patch
// the NaN case from one ballot
__device__ __forceinline__ double wave_nanmax(double v) {
    const bool has_nan = __any(v != v) != 0;
    v = fmax(v, dpp_move<DPP_XOR1>(v));
    v = fmax(v, dpp_move<DPP_XOR2>(v));
    v = fmax(v, dpp_move<DPP_HALF_MIRROR>(v));
    v = fmax(v, dpp_move<DPP_MIRROR>(v));
    v = fmax(v, dpp_bcast<DPP_BCAST15, 0xa>(v, v));
    v = fmax(v, dpp_bcast<DPP_BCAST31, 0xc>(v, v));
    const double m = lane_bcast(v, 63);
    return has_nan ? __builtin_nan("") : m;
}
// NS sums (s[]) and NM NaN-propagating maxima (m[]) at once: wave_sum()'s / wave_nanmax()'s steps, step by step on every value
template <int NS, int NM>
__device__ __forceinline__ void wave_reduce_n(double (&s)[NS], double (&m)[NM]) {
    bool has_nan[NM];
#pragma unroll
    for (int q = 0; q < NM; ++q) has_nan[q] = __any(m[q] != m[q]) != 0;
#define STM_WR_STEP(CTRL)                                                                                          \
    {                                                                                                              \
        _Pragma("unroll") for (int q = 0; q < NS; ++q) { const double t = dpp_move<CTRL>(s[q]); s[q] += t; }        \
        _Pragma("unroll") for (int q = 0; q < NM; ++q) { const double t = dpp_move<CTRL>(m[q]); m[q] = fmax(m[q], t); } \
        _Pragma("unroll") for (int q = 0; q < NS; ++q) asm volatile("" : "+v"(s[q]));                               \
        _Pragma("unroll") for (int q = 0; q < NM; ++q) asm volatile("" : "+v"(m[q]));                               \
    }
#define STM_WR_BC(CTRL, ROWS)                                                                                      \
    {                                                                                                              \
        _Pragma("unroll") for (int q = 0; q < NS; ++q) { const double t = dpp_bcast<CTRL, ROWS>(s[q], 0.0); s[q] += t; }   \
        _Pragma("unroll") for (int q = 0; q < NM; ++q) { const double t = dpp_bcast<CTRL, ROWS>(m[q], m[q]); m[q] = fmax(m[q], t); } \
        _Pragma("unroll") for (int q = 0; q < NS; ++q) asm volatile("" : "+v"(s[q]));                               \
        _Pragma("unroll") for (int q = 0; q < NM; ++q) asm volatile("" : "+v"(m[q]));                               \
    }
    STM_WR_STEP(DPP_XOR1)
    STM_WR_STEP(DPP_XOR2)
    STM_WR_STEP(DPP_HALF_MIRROR)
    STM_WR_STEP(DPP_MIRROR)
    STM_WR_BC(DPP_BCAST15, 0xa)
    STM_WR_BC(DPP_BCAST31, 0xc)
#undef STM_WR_STEP
#undef STM_WR_BC
#pragma unroll
    for (int q = 0; q < NS; ++q) s[q] = lane_bcast(s[q], 63);
#pragma unroll
    for (int q = 0; q < NM; ++q) { const double r = lane_bcast(m[q], 63); m[q] = has_nan[q] ? __builtin_nan("") : r; }
}
__device__ __forceinline__ bool wave_all(bool p) { return __all(p) != 0; }
// sqrt(d) and 1 / sqrt(d) together (d > 0): the coupled Goldschmidt iteration the compiler itself expands sqrt() into
// carries h ~ 1 / (2 sqrt(d)) along, so the reciprocal costs one more multiplication instead of an IEEE division.
// Both results are within ~1 ulp; arguments outside the normal range go through sqrt() and the division.
__device__ __forceinline__ void sqrt_and_rsqrt(double d, double &s, double &r) {
    if (!(d > 1e-280 && d < 1e280)) { s = sqrt(d); r = 1.0 / s; return; }
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    const double rr = fma(-g, g, d);
    g = fma(rr, h, g);
    e = fma(-h, g, 0.5);          // one more step on h alone: it is returned
    h = fma(h, e, h);
    s = g; r = h + h;
}

__device__ __forceinline__ bool wave_any(bool p) { return __any(p) != 0; }

// The same pair for a Cholesky pivot whose range the caller has checked (every pivot that passes lies in (32 eps, 1] x its
// diagonal entry, and the diagonal is tested once per attempt): no range test per pivot, and two coupled Goldschmidt steps
// without the final corrections -- both results within ~2 ulp (the seed is good to 2^-23: 2^-46, then rounding), which is
// what LAPACK's dpotf2 gets from sqrt and a division.
__device__ __forceinline__ void sqrt_and_rsqrt_pivot(double d, double &s, double &r) {
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    s = g; r = h + h;
}

// ---- Python / numpy scalar semantics scipy's line searches rely on ----------------
// builtin max/min keep the first argument unless a later one compares greater/less
__device__ __forceinline__ double py_max2(double a, double b) { return (b > a) ? b : a; }
__device__ __forceinline__ double py_max3(double a, double b, double c) { return py_max2(py_max2(a, b), c); }
__device__ __forceinline__ double py_min2(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double np_clip(double x, double lo, double hi) {
    if (x != x) return x;
    double r = x < lo ? lo : x;
    return r > hi ? hi : r;
}
__device__ __forceinline__ double np_sign(double x) {
    if (x != x) return x;
    return (double)((x > 0) - (x < 0));
}
__device__ __forceinline__ bool finite_d(double x) { return isfinite(x); }

// N IEEE divisions num[q] / den[q] at once: the code generator's own expansion of an fp64 `/` (v_div_scale x 2, v_rcp, the refinement
// of the reciprocal, v_div_fmas, v_div_fixup -- AMDGPUTargetLowering::LowerFDIV64), instruction for instruction, but stage by stage
// over all N quotients: every result has the bits `num / den` has (tests/test_gpu_parity.py::test_interleaved_division...), and the N
// dependent chains of ten instructions fill each other's latencies -- left to the compiler they run one after the other (DCSRCH's
// dcstep makes up to seven divisions per call, three of them independent of each other at two points).
template <int N>
__device__ __forceinline__ void div_n(const double (&num)[N], const double (&den)[N], double (&out)[N]) {
    double d0[N], r[N], e[N], n1[N], m[N], f4[N];
    bool flag[N], unused;
#define STM_DV(stmt) _Pragma("unroll") for (int q = 0; q < N; ++q) { stmt; }
#define STM_DP(a) _Pragma("unroll") for (int q = 0; q < N; ++q) asm volatile("" : "+v"(a[q]));
    STM_DV(d0[q] = __builtin_amdgcn_div_scale(num[q], den[q], false, &unused)) STM_DP(d0)
    STM_DV(r[q] = __builtin_amdgcn_rcp(d0[q])) STM_DP(r)
    STM_DV(e[q] = fma(-d0[q], r[q], 1.0)) STM_DP(e)
    STM_DV(r[q] = fma(r[q], e[q], r[q])) STM_DP(r)
    STM_DV(e[q] = fma(-d0[q], r[q], 1.0)) STM_DP(e)
    STM_DV(n1[q] = __builtin_amdgcn_div_scale(num[q], den[q], true, &flag[q])) STM_DP(n1)
    STM_DV(r[q] = fma(r[q], e[q], r[q])) STM_DP(r)
    STM_DV(m[q] = n1[q] * r[q]) STM_DP(m)
    STM_DV(f4[q] = fma(-d0[q], m[q], n1[q])) STM_DP(f4)
    STM_DV(f4[q] = __builtin_amdgcn_div_fmas(f4[q], r[q], m[q], flag[q])) STM_DP(f4)
    STM_DV(out[q] = __builtin_amdgcn_div_fixup(f4[q], den[q], num[q]))
#undef STM_DV
#undef STM_DP
}

// ---- natural logarithm, fdlibm e_log.c scheme (error < 1 ulp; 0.83 ulp measured over 2e7
// arguments against a long-double reference): x = 2^k (1+f), sqrt(1/2) < 1+f < sqrt(2),
// s = f/(2+f), log(1+f) = f - (f^2/2 - s (f^2/2 + R(s^2))).  About 45 VALU instructions against
// ~95 for the library routine; the objective takes one per word per evaluation.
__device__ __forceinline__ double log_pos(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1); denormals handled by the instruction
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0;
    const double d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    double s = f * r;
    s = fma(fma(-d, s, f), r, s);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    double res = dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    res = (x == 0.0) ? -INFINITY : res;
    res = (x == INFINITY) ? x : res;
    res = (x < 0.0) ? __builtin_nan("") : res;
    return res;  // NaN in -> NaN out through the arithmetic
}
// Two logarithms at once: log_pos()'s operations on both arguments, statement by statement -- each result has log_pos()'s bits, and
// the two dependent chains (~45 instructions each) fill each other's latencies instead of running one after the other (the solver's
// waves are latency-bound: a dependent fp64 instruction issues every ~9 cycles, an independent one every 4).
__device__ __forceinline__ void log_pos2(const double (&x)[2], double (&out)[2]) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    double m[2], f[2], d[2], r[2], s[2], z[2], w[2], t1[2], t2[2], R[2], hfsq[2], dk[2], res[2];
    int e[2];
    bool low[2];
    // (a stage is pinned for both arguments before the next one starts: left alone, the scheduler runs one chain after the other)
#define STM_L2(stmt) { constexpr int q = 0; stmt; } { constexpr int q = 1; stmt; }
#define STM_P2(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]));
    STM_L2(m[q] = __builtin_amdgcn_frexp_mant(x[q]))
    STM_L2(e[q] = __builtin_amdgcn_frexp_exp(x[q]))
    STM_L2(low[q] = m[q] < 0.70710678118654752440)
    STM_L2(m[q] = low[q] ? m[q] + m[q] : m[q])
    STM_L2(e[q] = low[q] ? e[q] - 1 : e[q])
    STM_L2(f[q] = m[q] - 1.0)
    STM_L2(d[q] = 2.0 + f[q])
    STM_L2(r[q] = __builtin_amdgcn_rcp(d[q]))
    STM_P2(r)
    double t[2];
    STM_L2(t[q] = fma(-d[q], r[q], 1.0)) STM_P2(t)
    STM_L2(r[q] = fma(r[q], t[q], r[q])) STM_P2(r)
    STM_L2(t[q] = fma(-d[q], r[q], 1.0)) STM_P2(t)
    STM_L2(r[q] = fma(r[q], t[q], r[q])) STM_P2(r)
    STM_L2(s[q] = f[q] * r[q]) STM_P2(s)
    STM_L2(t[q] = fma(-d[q], s[q], f[q])) STM_P2(t)
    STM_L2(s[q] = fma(t[q], r[q], s[q])) STM_P2(s)
    STM_L2(z[q] = s[q] * s[q]) STM_P2(z)
    STM_L2(w[q] = z[q] * z[q]) STM_P2(w)
    // the two polynomials of one argument are independent chains as well: four chains advance together
    double u1[2], u2[2];
    STM_L2(u1[q] = fma(w[q], Lg6, Lg4)) STM_L2(u2[q] = fma(w[q], Lg7, Lg5)) STM_P2(u1) STM_P2(u2)
    STM_L2(u1[q] = fma(w[q], u1[q], Lg2)) STM_L2(u2[q] = fma(w[q], u2[q], Lg3)) STM_P2(u1) STM_P2(u2)
    STM_L2(t1[q] = w[q] * u1[q]) STM_L2(u2[q] = fma(w[q], u2[q], Lg1)) STM_P2(t1) STM_P2(u2)
    STM_L2(t2[q] = z[q] * u2[q]) STM_P2(t2)
    STM_L2(R[q] = t2[q] + t1[q])
    STM_L2(hfsq[q] = 0.5 * f[q] * f[q])
    STM_L2(dk[q] = (double)e[q])
    STM_L2(res[q] = dk[q] * ln2_hi - ((hfsq[q] - (s[q] * (hfsq[q] + R[q]) + dk[q] * ln2_lo)) - f[q]))
    STM_L2(res[q] = (x[q] == 0.0) ? -INFINITY : res[q])
    STM_L2(res[q] = (x[q] == INFINITY) ? x[q] : res[q])
    STM_L2(out[q] = (x[q] < 0.0) ? __builtin_nan("") : res[q])
#undef STM_L2
#undef STM_P2
}
// log1p for x >= 0: log(u) + (x - (u - 1)) / u with u = 1 + x (the rounding error of u is
// recovered exactly by the second term)
__device__ __forceinline__ double log1p_pos(double x) {
    const double u = 1.0 + x;
    const double c = x - (u - 1.0);
    const double l = log_pos(u);
    return (u == INFINITY) ? l : l + c * __builtin_amdgcn_rcp(u);
}

// explicit waits (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8]) as builtins, so that the compiler's
// own counter bookkeeping sees them
__device__ __forceinline__ void wait_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0)
__device__ __forceinline__ void wait_lds() { __builtin_amdgcn_s_waitcnt(0xC07F); }    // lgkmcnt(0)
template <int N> __device__ __forceinline__ void wait_vmem_but() {   // vmcnt(N): all but the N youngest memory operations
    static_assert(N >= 0 && N < 64, "vmcnt is a six-bit counter");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
// ties a value to its place among the memory operations (a pure computation is otherwise free to be selected anywhere)
__device__ __forceinline__ void pin(double &v) { asm volatile("" : "+v"(v) :: "memory"); }

// LDS byte address of a pointer into the workgroup's LDS (what M0 carries for the LDS-DMA loads)
__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
// global_load_lds_dwordx4: every enabled lane (`mask`) moves 16 bytes from base + off to LDS byte address dst + 16 * lane.
// Hand-written, so that the compiler does not take it for a memory operation its LDS reads must wait for -- the caller
// waits (wait_vmem) before it reads what was fetched.  M0 and exec are saved and restored: the compiler owns both.
__device__ __forceinline__ void lds_dma16(const void *base, unsigned off, unsigned dst, unsigned long long mask) {
    unsigned keep;
    unsigned long long ex;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %3\n\ts_mov_b64 exec, %5\n\t"
                 "global_load_lds_dwordx4 %2, %4\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(ex) : "v"(off), "s"(dst), "s"(base), "s"(mask) : "memory");
}

// A prefetch into the L2 with no register destination: global_load_lds_dword moves 4 bytes per enabled lane from the lane's own
// 64-bit address to LDS byte address dst + 4 * lane (a dump nobody reads).  Nothing waits for it; a later vmcnt wait of the wave
// merely counts it.  (The caller's branch around the call is the lane mask.)
__device__ __forceinline__ void l2_prefetch_line(const void *addr, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(addr), "s"(dst) : "memory");
}

}  // namespace stm
