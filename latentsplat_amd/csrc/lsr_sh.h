// lsr_sh.h — real spherical-harmonics basis (degree <= 4) in the rasterizer's axis convention.
// Constants: reference src/misc/sh_utils.py:10-39.  Degrees 0-3 follow the published 3DGS
// rasterizer; degree 4 is the standard l=4 band, i.e. sh_utils.py:87-96 re-expressed with
// (x,y,z)_sh_utils = (y,z,x)_here.
#pragma once
#include <hip/hip_runtime.h>

namespace lsr {

// MAXDEG: compile-time bound on `deg` (lets the compiler drop the higher bands and their registers)
template <int MAXDEG = 4>
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float *b) {
    b[0] = 0.28209479177387814f;
    if (MAXDEG < 1 || deg < 1) return;
    const float C1 = 0.4886025119029199f;
    b[1] = -C1 * y;
    b[2] = C1 * z;
    b[3] = -C1 * x;
    if (MAXDEG < 2 || deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = 1.0925484305920792f * xy;
    b[5] = -1.0925484305920792f * yz;
    b[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
    b[7] = -1.0925484305920792f * xz;
    b[8] = 0.5462742152960396f * (xx - yy);
    if (MAXDEG < 3 || deg < 3) return;
    b[9] = -0.5900435899266435f * y * (3.0f * xx - yy);
    b[10] = 2.890611442640554f * xy * z;
    b[11] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
    b[12] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    b[13] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
    b[14] = 1.445305721320277f * z * (xx - yy);
    b[15] = -0.5900435899266435f * x * (xx - 3.0f * yy);
    if (MAXDEG < 4 || deg < 4) return;
    b[16] = 2.5033429417967046f * xy * (xx - yy);
    b[17] = -1.7701307697799304f * yz * (3.0f * xx - yy);
    b[18] = 0.9461746957575601f * xy * (7.0f * zz - 1.0f);
    b[19] = -0.6690465435572892f * yz * (7.0f * zz - 3.0f);
    b[20] = 0.10578554691520431f * (zz * (35.0f * zz - 30.0f) + 3.0f);
    b[21] = -0.6690465435572892f * xz * (7.0f * zz - 3.0f);
    b[22] = 0.47308734787878004f * (xx - yy) * (7.0f * zz - 1.0f);
    b[23] = -1.7701307697799304f * xz * (xx - 3.0f * yy);
    b[24] = 0.6258357354491761f * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
}

// d b[k] / d (x,y,z), treating x,y,z as independent.
// Only rows k < (MAXDEG+1)^2 of g are touched.
template <int MAXDEG = 4>
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float (*g)[3]) {
#pragma unroll
    for (int k = 0; k < (MAXDEG + 1) * (MAXDEG + 1); ++k) g[k][0] = g[k][1] = g[k][2] = 0.0f;
    if (MAXDEG < 1 || deg < 1) return;
    const float C1 = 0.4886025119029199f;
    g[1][1] = -C1; g[2][2] = C1; g[3][0] = -C1;
    if (MAXDEG < 2 || deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    const float a2 = 1.0925484305920792f, b2 = 0.31539156525252005f, c2 = 0.5462742152960396f;
    g[4][0] = a2 * y;   g[4][1] = a2 * x;
    g[5][1] = -a2 * z;  g[5][2] = -a2 * y;
    g[6][0] = -2.0f * b2 * x; g[6][1] = -2.0f * b2 * y; g[6][2] = 4.0f * b2 * z;
    g[7][0] = -a2 * z;  g[7][2] = -a2 * x;
    g[8][0] = 2.0f * c2 * x;  g[8][1] = -2.0f * c2 * y;
    if (MAXDEG < 3 || deg < 3) return;
    const float a3 = -0.5900435899266435f, b3 = 2.890611442640554f, c3 = -0.4570457994644658f,
                d3 = 0.3731763325901154f, e3 = 1.445305721320277f;
    g[9][0] = a3 * 6.0f * xy;                  g[9][1] = a3 * 3.0f * (xx - yy);
    g[10][0] = b3 * yz;  g[10][1] = b3 * xz;   g[10][2] = b3 * xy;
    g[11][0] = c3 * -2.0f * xy;  g[11][1] = c3 * (4.0f * zz - xx - 3.0f * yy);  g[11][2] = c3 * 8.0f * yz;
    g[12][0] = d3 * -6.0f * xz;  g[12][1] = d3 * -6.0f * yz;  g[12][2] = d3 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
    g[13][0] = c3 * (4.0f * zz - 3.0f * xx - yy);  g[13][1] = c3 * -2.0f * xy;  g[13][2] = c3 * 8.0f * xz;
    g[14][0] = e3 * 2.0f * xz;   g[14][1] = e3 * -2.0f * yz;  g[14][2] = e3 * (xx - yy);
    g[15][0] = a3 * 3.0f * (xx - yy);  g[15][1] = a3 * -6.0f * xy;
    if (MAXDEG < 4 || deg < 4) return;
    const float k0 = 2.5033429417967046f, k1 = -1.7701307697799304f, k2 = 0.9461746957575601f,
                k3 = -0.6690465435572892f, k4 = 0.10578554691520431f, k6 = 0.47308734787878004f,
                k8 = 0.6258357354491761f;
    g[16][0] = k0 * y * (3.0f * xx - yy);          g[16][1] = k0 * x * (xx - 3.0f * yy);
    g[17][0] = k1 * 6.0f * xy * z;  g[17][1] = k1 * 3.0f * z * (xx - yy);  g[17][2] = k1 * y * (3.0f * xx - yy);
    g[18][0] = k2 * y * (7.0f * zz - 1.0f);  g[18][1] = k2 * x * (7.0f * zz - 1.0f);  g[18][2] = k2 * 14.0f * xy * z;
    g[19][1] = k3 * z * (7.0f * zz - 3.0f);  g[19][2] = k3 * y * (21.0f * zz - 3.0f);
    g[20][2] = k4 * z * (140.0f * zz - 60.0f);
    g[21][0] = k3 * z * (7.0f * zz - 3.0f);  g[21][2] = k3 * x * (21.0f * zz - 3.0f);
    g[22][0] = k6 * 2.0f * x * (7.0f * zz - 1.0f);  g[22][1] = k6 * -2.0f * y * (7.0f * zz - 1.0f);  g[22][2] = k6 * 14.0f * z * (xx - yy);
    g[23][0] = k1 * 3.0f * z * (xx - yy);  g[23][1] = k1 * -6.0f * xy * z;  g[23][2] = k1 * x * (xx - 3.0f * yy);
    g[24][0] = k8 * 4.0f * x * (xx - 3.0f * yy);  g[24][1] = k8 * 4.0f * y * (yy - 3.0f * xx);
}

}  // namespace lsr
