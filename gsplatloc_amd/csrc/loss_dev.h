// The depth + Sobel-edge loss of one 16x16 pixel block with its gradient (device function shared by the stand-alone loss
// kernel, tracker.hip, and by the tiny-splat backward that computes its own upstream gradient, raster_px.hip).
//   loss  = depth_w * mean|d*m - g*m| + edge_w * mean|S(d*m) - S(g*m)|,  m = (d != 0) (no gradient),
//   S(x)  = sqrt(gx^2 + gy^2 + 1e-6), gx/gy = 3x3 Sobel / 8 with replicate padding (kornia.filters.sobel)
// (/root/reference/src/my_gsplat/loss.py:10-59, weights of gs_trainer_total.py:145-150).
#pragma once
#include "gsloc_common.h"

namespace gsl {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct LossLds {
  float sa[20][21], sg[20][21];    // masked depth images at (y0 - 2 + yy, x0 - 2 + xx), borders replicated
  float swx[18][19], swy[18][19];  // weights at (y0 - 1 + yy, x0 - 1 + xx); 0 outside the owned rows / image
  float red[4][2];
};

// Block (x0, y0) of rows [h0,h1) (owned rows [r0,r1) plus the one-row halo whose depth the owned rows' Sobel reads), 256
// threads, thread tid <-> pixel (y0 + (tid >> 4), x0 + (tid & 15)).  The block's depth values with a two-pixel apron go
// through LDS once; the edge-loss weights wx = s*gx/S, wy = s*gy/S (s = sign(S_d - S_g)) of the block and its one-pixel
// ring are computed there, and every pixel then gathers the adjoint of the replicate-padded Sobel from its 3x3
// neighbourhood.  render is [H,W,D], depth = channel D-1.  Returns through g the gradient of the loss with respect to
// this thread's pixel's depth (has_pixel: the pixel lies in rows [h0,h1) of the image), through l1 / le this thread's
// share of (sum|a-b|, sum|Sa-Sb|) over the block's owned pixels.  Two barriers inside; every thread must call it.
__device__ __forceinline__ void loss_block(LossLds& L, const float* __restrict__ render, int D,
                                           const float* __restrict__ gt, int W, int H, int r0, int r1, int h1,
                                           float depth_w, float edge_w, float inv_P, int x0, int y0, int tid, float& g_out,
                                           bool& has_pixel, float& l1_out, float& le_out) {
  auto& sa = L.sa;
  auto& sg = L.sg;
  auto& swx = L.swx;
  auto& swy = L.swy;
  for (int e = tid; e < 400; e += 256) {
    int yy = e / 20, xx = e - yy * 20;
    int y = clampi(y0 - 2 + yy, 0, H - 1), x = clampi(x0 - 2 + xx, 0, W - 1);
    float d = render[((size_t)y * W + x) * D + (D - 1)];
    sa[yy][xx] = d;                                          // d * m == d
    sg[yy][xx] = d != 0.f ? gt[(size_t)y * W + x] : 0.f;    // g * m
  }
  __syncthreads();
  float l1 = 0.f, le = 0.f;
  for (int e = tid; e < 324; e += 256) {
    int yy = e / 18, xx = e - yy * 18;
    int y = y0 - 1 + yy, x = x0 - 1 + xx;
    float wx = 0.f, wy = 0.f;
    if (y >= r0 && y < r1 && x >= 0 && x < W) {
      const int cy = yy + 1, cx = xx + 1;  // position in sa / sg
      float gxa = ((sa[cy - 1][cx + 1] - sa[cy - 1][cx - 1]) + 2.f * (sa[cy][cx + 1] - sa[cy][cx - 1]) +
                   (sa[cy + 1][cx + 1] - sa[cy + 1][cx - 1])) * 0.125f;
      float gya = ((sa[cy + 1][cx - 1] - sa[cy - 1][cx - 1]) + 2.f * (sa[cy + 1][cx] - sa[cy - 1][cx]) +
                   (sa[cy + 1][cx + 1] - sa[cy - 1][cx + 1])) * 0.125f;
      float gxb = ((sg[cy - 1][cx + 1] - sg[cy - 1][cx - 1]) + 2.f * (sg[cy][cx + 1] - sg[cy][cx - 1]) +
                   (sg[cy + 1][cx + 1] - sg[cy + 1][cx - 1])) * 0.125f;
      float gyb = ((sg[cy + 1][cx - 1] - sg[cy - 1][cx - 1]) + 2.f * (sg[cy + 1][cx] - sg[cy - 1][cx]) +
                   (sg[cy + 1][cx + 1] - sg[cy - 1][cx + 1])) * 0.125f;
      float Sa = sqrtf(gxa * gxa + gya * gya + 1e-6f), Sb = sqrtf(gxb * gxb + gyb * gyb + 1e-6f);
      float sgn = (Sa > Sb) ? 1.f : ((Sa < Sb) ? -1.f : 0.f);
      wx = sgn * gxa / Sa;
      wy = sgn * gya / Sa;
      if (yy >= 1 && yy <= 16 && xx >= 1 && xx <= 16 && y < h1) {  // a pixel of this block: its loss terms
        l1 += fabsf(sa[cy][cx] - sg[cy][cx]);
        le += fabsf(Sa - Sb);
      }
    }
    swx[yy][xx] = wx;
    swy[yy][xx] = wy;
  }
  __syncthreads();
  {
    int ly = tid >> 4, lx = tid & 15;
    int i = y0 + ly, j = x0 + lx;
    has_pixel = i < h1 && j < W;
    g_out = 0.f;
    if (has_pixel) {
      float d = sa[ly + 2][lx + 2];
      float g = 0.f;
      if (d != 0.f) {
        // Sobel taps: Kx[a][b] = (b-1)*(a==1?2:1)/8, Ky[a][b] = (a-1)*(b==1?2:1)/8 at offset (a-1, b-1); tap (a,b)
        // of pixel p reads clamp(p + (a-1, b-1)) and contributes to q when that equals q
        float acc = 0.f;
        if (i > 0 && i < H - 1 && j > 0 && j < W - 1) {
          // no tap is clamped onto a pixel that is not on the image border: plain transposed 3x3 correlation
          // (tap (a,b) of p = q - (a-1, b-1) reads q; weights outside the owned rows are zero in swx / swy)
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
              float kx = (float)(b - 1) * (a == 1 ? 2.f : 1.f) * 0.125f;
              float ky = (float)(a - 1) * (b == 1 ? 2.f : 1.f) * 0.125f;
              if (kx != 0.f) acc += swx[ly + 2 - a][lx + 2 - b] * kx;
              if (ky != 0.f) acc += swy[ly + 2 - a][lx + 2 - b] * ky;
            }
        } else
        for (int pi = max(i - 1, max(r0, 0)); pi <= min(i + 1, r1 - 1); ++pi)
          for (int pj = max(j - 1, 0); pj <= min(j + 1, W - 1); ++pj) {
            float wx = swx[pi - y0 + 1][pj - x0 + 1], wy = swy[pi - y0 + 1][pj - x0 + 1];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
              for (int b = 0; b < 3; ++b)
                if (clampi(pi + a - 1, 0, H - 1) == i && clampi(pj + b - 1, 0, W - 1) == j) {
                  float kx = (float)(b - 1) * (a == 1 ? 2.f : 1.f) * 0.125f;
                  float ky = (float)(a - 1) * (b == 1 ? 2.f : 1.f) * 0.125f;
                  acc += wx * kx + wy * ky;
                }
          }
        g = edge_w * acc;
        if (i >= r0 && i < r1) {
          float diff = d - sg[ly + 2][lx + 2];
          g += depth_w * ((diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f));
        }
        g *= inv_P;
      }
      g_out = g;
    }
  }
  l1_out = l1;
  le_out = le;
}

}  // namespace gsl
