// Which translation unit compiles which instantiations of lbm::k_fused.
//
// The multi-step sweep is one kernel template with some 360 instantiations (precision, nodes per
// lane, steps per launch, scheme, scene, layout, arithmetic), each a long compile.  The C-ABI file
// declares the families below `extern template` and launches them; lbm_fused_inst_*.hip define one
// family set each (LBM_FUSED_DEFINE_SET selects it), so the families compile side by side.  A family
// missing from these lists is still correct: it is then instantiated where it is launched.
//
// X(T, V, K, PUSH, LID, AOS, ARITH, NARROW) -- the template arguments of lbm::k_fused.
#pragma once

#include "lbm_device.h"
#include "lbm_fused.h"

#define LBM_FUSED_STEPS(X, T, V, PUSH, LID, AOS, ARITH)                                                  \
    X(T, V, 1, PUSH, LID, AOS, ARITH, false) X(T, V, 2, PUSH, LID, AOS, ARITH, false)                     \
    X(T, V, 3, PUSH, LID, AOS, ARITH, false) X(T, V, 4, PUSH, LID, AOS, ARITH, false)                     \
    X(T, V, 5, PUSH, LID, AOS, ARITH, false) X(T, V, 6, PUSH, LID, AOS, ARITH, false)                     \
    X(T, V, 7, PUSH, LID, AOS, ARITH, false) X(T, V, 8, PUSH, LID, AOS, ARITH, false)
// pull and push, periodic and cavity
#define LBM_FUSED_SWEEPS(X, T, V, AOS, ARITH)                                                            \
    LBM_FUSED_STEPS(X, T, V, false, false, AOS, ARITH) LBM_FUSED_STEPS(X, T, V, true, false, AOS, ARITH)  \
    LBM_FUSED_STEPS(X, T, V, false, true, AOS, ARITH) LBM_FUSED_STEPS(X, T, V, true, true, AOS, ARITH)
// periodic only (the rolled arithmetic), and the five-step kernels with halos of whole lanes
#define LBM_FUSED_PERIODIC(X, T, V, AOS, ARITH)                                                          \
    LBM_FUSED_STEPS(X, T, V, false, false, AOS, ARITH) LBM_FUSED_STEPS(X, T, V, true, false, AOS, ARITH)
#define LBM_FUSED_NARROW(X, T, V, ARITH)                                                                 \
    X(T, V, 5, false, false, false, ARITH, true) X(T, V, 5, true, false, false, ARITH, true)

// set 1: fp32, [9][NY][NX], strict and rolled arithmetic
#define LBM_FUSED_SET_1(X)                                                                               \
    LBM_FUSED_SWEEPS(X, float, 2, false, 0) LBM_FUSED_SWEEPS(X, float, 1, false, 0)                       \
    LBM_FUSED_PERIODIC(X, float, 2, false, 1) LBM_FUSED_NARROW(X, float, 2, 0) LBM_FUSED_NARROW(X, float, 2, 1)
// set 2: fp32, [9][NY][NX], contracted arithmetic
#define LBM_FUSED_SET_2(X) LBM_FUSED_SWEEPS(X, float, 2, false, 2) LBM_FUSED_NARROW(X, float, 2, 2)
// set 3: fp32, [NY][NX][9]
#define LBM_FUSED_SET_3(X)                                                                               \
    LBM_FUSED_SWEEPS(X, float, 2, true, 0) LBM_FUSED_SWEEPS(X, float, 1, true, 0) LBM_FUSED_SWEEPS(X, float, 2, true, 2)
// set 4: fp64, [9][NY][NX]
#define LBM_FUSED_SET_4(X)                                                                               \
    LBM_FUSED_SWEEPS(X, double, 1, false, 0) LBM_FUSED_PERIODIC(X, double, 1, false, 1)                   \
    LBM_FUSED_SWEEPS(X, double, 1, false, 2) LBM_FUSED_NARROW(X, double, 1, 0) LBM_FUSED_NARROW(X, double, 1, 1) \
    LBM_FUSED_NARROW(X, double, 1, 2)
// set 5: fp64, [NY][NX][9]
#define LBM_FUSED_SET_5(X) LBM_FUSED_SWEEPS(X, double, 1, true, 0) LBM_FUSED_SWEEPS(X, double, 1, true, 2)

#define LBM_FUSED_ALL_SETS(X) LBM_FUSED_SET_1(X) LBM_FUSED_SET_2(X) LBM_FUSED_SET_3(X) LBM_FUSED_SET_4(X) LBM_FUSED_SET_5(X)

#define LBM_FUSED_DECLARE(T, V, K, PUSH, LID, AOS, ARITH, NARROW)                                        \
    extern template __global__ void lbm::k_fused<T, V, K, PUSH, LID, AOS, ARITH, NARROW>(const T *, T *, const lbm::FusedGeom, const T, const T);
#define LBM_FUSED_DEFINE(T, V, K, PUSH, LID, AOS, ARITH, NARROW)                                         \
    template __global__ void lbm::k_fused<T, V, K, PUSH, LID, AOS, ARITH, NARROW>(const T *, T *, const lbm::FusedGeom, const T, const T);
