// internal interface between conv.hip (dispatch) and conv_small.hip (direct VALU kernels for tiny channel counts)
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"  // BnFold

struct Geom {
    int N, Hs, Ws, Cs, up, Ho, Wo, Cn, KH, KW, stride, pad;
};
typedef Geom SmallGeom;

bool conv_small_ok(const SmallGeom& g);
bool conv_small_wgrad_ok(const SmallGeom& g);
void conv_small_wgrad_plan(const SmallGeom& g, int& P, unsigned& chunk);
int conv_small_fwd(const SmallGeom& g, int nblocks, const float* x, const float* scale, const float* shift, int relu,
                   const float* wT, const float* bias, const float* res, float* y, double* partial, int CnPad,
                   const BnFold& fold, hipStream_t st);
int conv_small_dgrad(const SmallGeom& g, int nblocks, const float* gy, const float* wD, const float* x, const float* scale,
                     const float* shift, int relu, const float* mean, const float* invstd, float* gv, double* partial,
                     int CsPad, hipStream_t st);
int conv_small_wgrad(const SmallGeom& g, const float* x, const float* scale, const float* shift, int relu, const float* gy,
                     int has_bias, float* partial, hipStream_t st);
