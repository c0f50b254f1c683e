// internal interface between conv.hip (dispatch) and conv_tile.hip (image-tile MFMA convolution)
#pragma once
#include <hip/hip_runtime.h>
#include "conv_small.h"  // struct Geom

#define TILE_MAXT 16  // taps per (parity class of a) layer

struct TileTaps {
    int n;
    int off[TILE_MAXT];   // LDS float offset of the tap relative to a position's origin in the virtual grid
    int wrow[TILE_MAXT];  // tap index kh*KW + kw (x CK = first weight row)
};

struct TilePlan {
    int N;                  // images
    int srcH, srcW, CK, NC; // staged tensor (x, or gy for the data gradient) per image; output channels
    int Hv, Wv, CKp;        // virtual grid in LDS (rows, cols, padded channel stride)
    int voffy, voffx;       // source coordinate u = v - voff (then >> ush); inside iff 0 <= u < lim
    int ush, limH, limW;
    int rowsH, rowsW, rowsPI;  // output positions per image (per parity class)
    int rstride;            // virtual-grid step per output coordinate
    int childmode;          // data gradient of an up-sampling layer: rows are (source pixel << 2 | child)
    int s2;                 // data gradient of a stride-2 layer: blockIdx.z = input parity class
    int IPB;                // images per block
    int vfloats;            // IPB * Hv * Wv * CKp
    int tpc;                // taps per weight chunk
    int nt, rbw, cpad;      // column tiles per wave, row blocks per wave, padded channel count of the partial sums
    TileTaps taps[4];
};

bool conv_tile_plan(const Geom& g, int mode, TilePlan& pl, dim3& grid, size_t& smem);
int conv_tile_fwd(const TilePlan& pl, dim3 grid, size_t smem, hipStream_t st, const float* x, const float* scale,
                  const float* shift, int relu, const float* wT, const float* bias, const float* res, float* y,
                  double* partial, const BnFold& fold = BnFold{});
int conv_tile_dgrad(const TilePlan& pl, dim3 grid, size_t smem, hipStream_t st, const float* gy, const float* wD,
                    const float* x, const float* scale, const float* shift, int relu, const float* mean,
                    const float* invstd, float* gv, double* partial);
