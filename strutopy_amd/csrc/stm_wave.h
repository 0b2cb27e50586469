// stm_wave.h -- wave64 helpers and scalar semantics shared by the E-step kernels (gfx950).
//
// One wavefront (64 lanes) owns one document.  Vectors of length n = K-1 live one
// component per lane (VPL components per lane when n > 64); scalars of the line search
// are wave-uniform and are pinned to SGPRs with uni() so the solver's control flow
// compiles to scalar branches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stm {

constexpr int WAVE = 64;

__device__ __forceinline__ double uni(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

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
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // row_half_mirror
constexpr int DPP_MIRROR = 0x140;      // row_mirror

__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<DPP_XOR1>(v);
    v += dpp_move<DPP_XOR2>(v);
    v += dpp_move<DPP_HALF_MIRROR>(v);
    v += dpp_move<DPP_MIRROR>(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
// np.max semantics: NaN propagates
__device__ __forceinline__ double nanmax(double a, double b) {
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}
__device__ __forceinline__ double wave_nanmax(double v) {
    v = nanmax(v, dpp_move<DPP_XOR1>(v));
    v = nanmax(v, dpp_move<DPP_XOR2>(v));
    v = nanmax(v, dpp_move<DPP_HALF_MIRROR>(v));
    v = nanmax(v, dpp_move<DPP_MIRROR>(v));
    return nanmax(nanmax(lane_bcast(v, 0), lane_bcast(v, 16)), nanmax(lane_bcast(v, 32), lane_bcast(v, 48)));
}
__device__ __forceinline__ bool wave_all(bool p) { return __all(p) != 0; }
__device__ __forceinline__ bool wave_any(bool p) { return __any(p) != 0; }

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

}  // namespace stm
