// internal interface between conv.hip (dispatch) and conv_wtile.hip (image-tile MFMA weight gradient)
#pragma once
#include <hip/hip_runtime.h>
#include "conv_small.h"  // struct Geom

#define WT_MAXT 16  // taps (KH*KW)

struct WTilePlan {
    int N;                    // images
    int srcH, srcW, CK, Cn;   // x per image, input / output channels
    int Hv, Wv, CKp;          // virtual grid of act(x) in LDS
    int voffy, voffx, ush, limH, limW;
    int rowsH, rowsW, rowsPI, rowsPIp, rstride;  // output positions per image (padded to a multiple of 4)
    int vec4;                 // Cs % 4 == 0: float4 staging of x
    int gvec;                 // gy rows contiguous and Cn % 4 == 0: float4 staging of gy
    int K, Kp, has_bias;      // k-rows: taps * Cs (+ 1 bias row)
    int nkt, nn, ny, CnP;     // 16-row k tiles, 16-col n tiles per block, blocks along n, padded row length of G
    int IPB, vfloats, gfloats;
    int tapoff[WT_MAXT];      // LDS float offset of each tap (kh*KW + kw) inside the virtual grid, -1 = never touches
};

bool conv_wtile_plan(const Geom& g, int has_bias, WTilePlan& pl, int& nblocks, size_t& smem);
int conv_wtile_nparts(int nblocks);
int conv_wtile(const WTilePlan& pl, int nblocks, size_t smem, hipStream_t st, const float* x, const float* scale,
               const float* shift, int relu, const float* gy, float* partial);
