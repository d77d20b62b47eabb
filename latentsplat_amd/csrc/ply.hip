// ply.hip — 3DGS .ply export (include/lsr_ply.h): per-Gaussian vertex packing on the device and
// the binary file writer on the host.  Restates /root/reference/src/model/ply_export.py:26-92
// (no code taken); the quaternion <-> matrix conversions follow scipy's documented
// Rotation.from_quat / from_matrix (largest-of-diagonal-and-trace branch) in double precision so
// that the exported quaternion has the same sign as the reference's.
#include <stdio.h>

#include "lsr_internal.h"
#include "lsr_ply.h"

namespace lsr {

__global__ __launch_bounds__(256) void k_ply_pack(int64_t n, int d_sh, lsr_ply_inputs in, float *__restrict__ out) {
    __shared__ double Rv[9];                             // viewer rotation
    if (threadIdx.x == 0) {
        double m[9], inv[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) m[3 * r + c] = in.extrinsics[4 * r + c];
        const double A = m[4] * m[8] - m[5] * m[7], B = -(m[3] * m[8] - m[5] * m[6]), C = m[3] * m[7] - m[4] * m[6];
        const double id = 1.0 / (m[0] * A + m[1] * B + m[2] * C);
        inv[0] = A * id; inv[1] = -(m[1] * m[8] - m[2] * m[7]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
        inv[3] = B * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id;  inv[5] = -(m[0] * m[5] - m[2] * m[3]) * id;
        inv[6] = C * id; inv[7] = -(m[0] * m[7] - m[1] * m[6]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
        const double s = 0.70710678118654752440;
        // Rz(-45 deg) @ [[0,0,1],[-1,0,0],[0,-1,0]]
        const double base[9] = {-s, 0, s, -s, 0, -s, 0, -1, 0};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                Rv[3 * r + c] = base[3 * r] * inv[c] + base[3 * r + 1] * inv[3 + c] + base[3 * r + 2] * inv[6 + c];
    }
    __syncthreads();
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const float sf = in.scale_factor[0];
    float p[3];
    for (int k = 0; k < 3; ++k) p[k] = (in.means[3 * g + k] - in.center[k]) / sf;
    float *v = out + g * LSR_PLY_VERTEX_FLOATS;
    for (int r = 0; r < 3; ++r)
        v[r] = (float)Rv[3 * r] * p[0] + (float)Rv[3 * r + 1] * p[1] + (float)Rv[3 * r + 2] * p[2];
    v[3] = v[4] = v[5] = 0.0f;
    for (int c = 0; c < 3; ++c) v[6 + c] = in.harmonics[(3 * g + c) * (int64_t)d_sh];
    v[9] = in.opacities[g];
    for (int k = 0; k < 3; ++k) v[10 + k] = logf(in.scales[3 * g + k] / sf);
    // orientation: normalise, to matrix, rotate, back to a quaternion
    double x = in.rotations[4 * g], y = in.rotations[4 * g + 1], z = in.rotations[4 * g + 2], w = in.rotations[4 * g + 3];
    const double nq = 1.0 / sqrt(x * x + y * y + z * z + w * w);
    x *= nq; y *= nq; z *= nq; w *= nq;
    const double Q[9] = {x * x - y * y - z * z + w * w, 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), -x * x + y * y - z * z + w * w, 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), -x * x - y * y + z * z + w * w};
    double M[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M[3 * r + c] = Rv[3 * r] * Q[c] + Rv[3 * r + 1] * Q[3 + c] + Rv[3 * r + 2] * Q[6 + c];
    const double tr = M[0] + M[4] + M[8];
    const double dec[4] = {M[0], M[4], M[8], tr};
    int choice = 0;
    for (int k = 1; k < 4; ++k)
        if (dec[k] > dec[choice]) choice = k;
    double q[4];
    if (choice != 3) {
        const int i = choice, j = (i + 1) % 3, k = (j + 1) % 3;
        q[i] = 1.0 - tr + 2.0 * M[4 * i];
        q[j] = M[3 * j + i] + M[3 * i + j];
        q[k] = M[3 * k + i] + M[3 * i + k];
        q[3] = M[3 * k + j] - M[3 * j + k];
    } else {
        q[0] = M[7] - M[5];
        q[1] = M[2] - M[6];
        q[2] = M[3] - M[1];
        q[3] = 1.0 + tr;
    }
    const double qn = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    v[13] = (float)(q[3] * qn);
    v[14] = (float)(q[0] * qn);
    v[15] = (float)(q[1] * qn);
    v[16] = (float)(q[2] * qn);
}

}  // namespace lsr

using namespace lsr;

extern "C" {

int lsr_ply_pack(int64_t n, int32_t d_sh, const lsr_ply_inputs *in, float *vertices, lsr_stream_t stream) {
    note_hip_error(0);
    if (n < 0 || d_sh < 1) return LSR_EINVAL;
    if (!in) return LSR_ENULL;
    if (n == 0) return LSR_OK;
    if (!in->extrinsics || !in->means || !in->scales || !in->rotations || !in->harmonics || !in->opacities ||
        !in->center || !in->scale_factor || !vertices)
        return LSR_ENULL;
    if ((n + 255) / 256 > 0x7fffffffLL) return LSR_EUNSUPPORTED;
    hipLaunchKernelGGL(k_ply_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, d_sh, *in, vertices);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { note_hip_error((int)e); return LSR_ELAUNCH; }
    return LSR_OK;
}

int lsr_ply_write_host(const char *path, const float *vertices_host, int64_t n) {
    if (!path || (n > 0 && !vertices_host)) return LSR_ENULL;
    if (n < 0) return LSR_EINVAL;
    FILE *f = fopen(path, "wb");
    if (!f) return LSR_EINVAL;
    static const char *props[LSR_PLY_VERTEX_FLOATS] = {"x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2",
                                                       "opacity", "scale_0", "scale_1", "scale_2",
                                                       "rot_0", "rot_1", "rot_2", "rot_3"};
    bool ok = fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %lld\n", (long long)n) > 0;
    for (int k = 0; k < LSR_PLY_VERTEX_FLOATS && ok; ++k) ok = fprintf(f, "property float %s\n", props[k]) > 0;
    ok = ok && fprintf(f, "end_header\n") > 0;
    if (ok && n > 0) ok = fwrite(vertices_host, sizeof(float) * LSR_PLY_VERTEX_FLOATS, (size_t)n, f) == (size_t)n;
    ok = (fclose(f) == 0) && ok;
    return ok ? LSR_OK : LSR_EINVAL;
}

}  // extern "C"
