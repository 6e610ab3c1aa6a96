// Correlation geometry (ops/correlation_op.h:36-51), shared by the correlation kernels.
#pragma once
#include <math.h>

struct CorrGeom {
  int k, kr, md, pad, s1, s2;
  int r, gw;       // displacement grid radius / width
  int oh, ow, oc;  // output size
};

static inline CorrGeom make_corr_geom(int H, int W, int k, int md, int pad, int s1, int s2) {
  CorrGeom g;
  g.k = k; g.kr = (k - 1) / 2; g.md = md; g.pad = pad; g.s1 = s1; g.s2 = s2;
  const int ph = H + 2 * pad, pw = W + 2 * pad, border = md + g.kr;
  g.r = md / s2;
  g.gw = 2 * g.r + 1;
  g.ow = (int)ceilf((float)(pw - 2 * border) / (float)s1);
  g.oh = (int)ceilf((float)(ph - 2 * border) / (float)s1);
  g.oc = g.gw * g.gw;
  return g;
}
