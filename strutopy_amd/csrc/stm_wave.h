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

// xor-butterfly all-reduce: every lane ends with the bitwise-identical result
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return uni(v);
}
// np.max semantics: NaN propagates
__device__ __forceinline__ double nanmax(double a, double b) {
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}
__device__ __forceinline__ double wave_nanmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = nanmax(v, __shfl_xor(v, o));
    return uni(v);
}
__device__ __forceinline__ bool wave_all(bool p) { return __all(p) != 0; }
__device__ __forceinline__ bool wave_any(bool p) { return __any(p) != 0; }

// value held by lane `src` (uniform src) -> uniform
__device__ __forceinline__ double lane_bcast(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
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

}  // namespace stm
