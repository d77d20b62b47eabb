// adapter.hip — the "Gaussian adapter tail" (include/lsr_adapter.h): raw network outputs ->
// means / covariances / scales / rotations, and its backward, one launch each.
//
// Reference behaviour restated (no code taken): /root/reference/src/model/encoder/common/
// gaussian_adapter.py:63-114 (forward), :116-127 (get_scale_multiplier), gaussians.py:8-44
// (quaternion_to_matrix xyzw, build_covariance), src/geometry/projection.py:74-114 (unproject,
// get_world_rays).  The reference runs this as ~45 PyTorch ops on (b,v,r,srf,spp) tensors.
//
// Decomposition: thread per parameter row (camera, ray); the row's raw scale / quaternion /
// coordinate are shared by its `samples` depth samples, so the quaternion->matrix work and the
// gradient sums over samples stay in registers.  Per-camera constants (c2w rotation, origin,
// K^-1, scale multiplier) are computed once per block in LDS (inverse in double).  HBM-bound:
// forward reads 4*(7 + 2 + S) and writes 4*(S*(3+cov+3) + 4) bytes per row.
#include "lsr_internal.h"
#include "lsr_adapter.h"

namespace lsr {

struct AdapterCam {
    float C[9];      // c2w rotation, row-major
    float org[3];    // camera origin
    float Ki[9];     // inverse normalised intrinsics
    float mult;      // get_scale_multiplier
};

__device__ static void adapter_camera(const float *__restrict__ E, const float *__restrict__ K, int height,
                                      int width, AdapterCam &cam) {
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) cam.C[3 * r + c] = E[4 * r + c];
        cam.org[r] = E[4 * r + 3];
    }
    const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
    const double A = e * i - f * h, B = -(d * i - f * g), Cc = d * h - e * g;
    const double id = 1.0 / (a * A + b * B + c * Cc);
    cam.Ki[0] = (float)(A * id);  cam.Ki[1] = (float)(-(b * i - c * h) * id); cam.Ki[2] = (float)((b * f - c * e) * id);
    cam.Ki[3] = (float)(B * id);  cam.Ki[4] = (float)((a * i - c * g) * id);  cam.Ki[5] = (float)(-(a * f - c * d) * id);
    cam.Ki[6] = (float)(Cc * id); cam.Ki[7] = (float)(-(a * h - b * g) * id); cam.Ki[8] = (float)((a * e - b * d) * id);
    // 0.1 * sum(K[:2,:2]^-1 @ (1/w, 1/h))
    const double det2 = a * e - b * d;
    const double px = 1.0 / (double)width, py = 1.0 / (double)height;
    const double mx = (e * px - b * py) / det2, my = (-d * px + a * py) / det2;
    cam.mult = (float)(0.1 * mx + 0.1 * my);
}

struct AdapterRow {
    float sig[3], base[3];   // sigmoid(raw scale), min + (max-min)*sigmoid
    float q[4];              // normalised quaternion xyzw
    float len, u;            // |raw q|, 1/(|raw q| + eps)
    float n, t;              // q.q + eps, 2/n
    float A[9];              // c2w_rot @ R(q), row-major
    float vlen;              // |K^-1 [x y 1]|
    float dc[3], dw[3];      // camera- and world-space unit ray direction
};

__device__ static void adapter_row(const AdapterCam &cam, const float *__restrict__ raw,
                                   const float *__restrict__ xy, float smin, float smax, float eps,
                                   AdapterRow &r) {
    for (int k = 0; k < 3; ++k) {
        r.sig[k] = 1.0f / (1.0f + __expf(-raw[k]));
        r.base[k] = smin + (smax - smin) * r.sig[k];
    }
    const float rx = raw[3], ry = raw[4], rz = raw[5], rw = raw[6];
    r.len = sqrtf(rx * rx + ry * ry + rz * rz + rw * rw);
    r.u = 1.0f / (r.len + eps);
    const float i = rx * r.u, j = ry * r.u, k = rz * r.u, w = rw * r.u;
    r.q[0] = i; r.q[1] = j; r.q[2] = k; r.q[3] = w;
    r.n = i * i + j * j + k * k + w * w + 1e-8f;   // quaternion_to_matrix's own eps (gaussians.py:11)
    r.t = 2.0f / r.n;
    const float t = r.t;
    const float R[9] = {1.0f - t * (j * j + k * k), t * (i * j - k * w), t * (i * k + j * w),
                        t * (i * j + k * w), 1.0f - t * (i * i + k * k), t * (j * k - i * w),
                        t * (i * k - j * w), t * (j * k + i * w), 1.0f - t * (i * i + j * j)};
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            r.A[3 * a + b] = cam.C[3 * a] * R[b] + cam.C[3 * a + 1] * R[3 + b] + cam.C[3 * a + 2] * R[6 + b];
    const float x = xy[0], y = xy[1];
    const float v0 = cam.Ki[0] * x + cam.Ki[1] * y + cam.Ki[2];
    const float v1 = cam.Ki[3] * x + cam.Ki[4] * y + cam.Ki[5];
    const float v2 = cam.Ki[6] * x + cam.Ki[7] * y + cam.Ki[8];
    r.vlen = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
    const float iv = 1.0f / r.vlen;
    r.dc[0] = v0 * iv; r.dc[1] = v1 * iv; r.dc[2] = v2 * iv;
    for (int a = 0; a < 3; ++a)
        r.dw[a] = cam.C[3 * a] * r.dc[0] + cam.C[3 * a + 1] * r.dc[1] + cam.C[3 * a + 2] * r.dc[2];
}

constexpr int kAdapterThreads = 256;

template <int COV>
__global__ __launch_bounds__(kAdapterThreads) void k_adapter_fwd(lsr_adapter_dims d, lsr_adapter_inputs in,
                                                                  lsr_adapter_outputs out) {
    __shared__ AdapterCam cam;
    const int c = blockIdx.y;
    if (threadIdx.x == 0) adapter_camera(in.extrinsics + 16 * (size_t)c, in.intrinsics + 9 * (size_t)c, d.height, d.width, cam);
    __syncthreads();
    const int ray = blockIdx.x * kAdapterThreads + threadIdx.x;
    if (ray >= d.rays) return;
    const size_t row = (size_t)c * d.rays + ray;
    AdapterRow r;
    adapter_row(cam, in.raw + row * (size_t)d.raw_stride, in.coordinates + 2 * row, d.scale_min, d.scale_max, d.eps, r);
    reinterpret_cast<float4 *>(out.rotations)[row] = make_float4(r.q[0], r.q[1], r.q[2], r.q[3]);
    for (int s = 0; s < d.samples; ++s) {
        const size_t e = row * d.samples + s;
        const float depth = in.depths[e];
        const float dm = depth * cam.mult;
        float sc[3], D[3];
        for (int k = 0; k < 3; ++k) {
            sc[k] = r.base[k] * dm;
            D[k] = sc[k] * sc[k];
        }
        float S[9];
        for (int a = 0; a < 3; ++a)
            for (int b = a; b < 3; ++b)
                S[3 * a + b] = r.A[3 * a] * D[0] * r.A[3 * b] + r.A[3 * a + 1] * D[1] * r.A[3 * b + 1] +
                               r.A[3 * a + 2] * D[2] * r.A[3 * b + 2];
        float *cv = out.covariances + e * COV;
        if (COV == 9) {
            cv[0] = S[0]; cv[1] = S[1]; cv[2] = S[2];
            cv[3] = S[1]; cv[4] = S[4]; cv[5] = S[5];
            cv[6] = S[2]; cv[7] = S[5]; cv[8] = S[8];
        } else {
            cv[0] = S[0]; cv[1] = S[1]; cv[2] = S[2]; cv[3] = S[4]; cv[4] = S[5]; cv[5] = S[8];
        }
        for (int a = 0; a < 3; ++a) {
            out.means[3 * e + a] = cam.org[a] + r.dw[a] * depth;
            out.scales[3 * e + a] = sc[a];
        }
    }
}

template <int COV>
__global__ __launch_bounds__(kAdapterThreads) void k_adapter_bwd(lsr_adapter_dims d, lsr_adapter_inputs in,
                                                                  lsr_adapter_out_grads g, lsr_adapter_in_grads o) {
    __shared__ AdapterCam cam;
    const int c = blockIdx.y;
    if (threadIdx.x == 0) adapter_camera(in.extrinsics + 16 * (size_t)c, in.intrinsics + 9 * (size_t)c, d.height, d.width, cam);
    __syncthreads();
    const int ray = blockIdx.x * kAdapterThreads + threadIdx.x;
    if (ray >= d.rays) return;
    const size_t row = (size_t)c * d.rays + ray;
    AdapterRow r;
    adapter_row(cam, in.raw + row * (size_t)d.raw_stride, in.coordinates + 2 * row, d.scale_min, d.scale_max, d.eps, r);

    float dA[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // d/d(c2w_rot @ R)
    float dbase[3] = {0, 0, 0};
    float ddw[3] = {0, 0, 0};                     // d/d(world ray direction)
    for (int s = 0; s < d.samples; ++s) {
        const size_t e = row * d.samples + s;
        const float depth = in.depths[e];
        const float dm = depth * cam.mult;
        const float *gc = g.covariances + e * COV;
        float G[9];
        if (COV == 9) {
            for (int k = 0; k < 9; ++k) G[k] = gc[k];
        } else {
            G[0] = gc[0]; G[1] = gc[1]; G[2] = gc[2];
            G[3] = 0.0f;  G[4] = gc[3]; G[5] = gc[4];
            G[6] = 0.0f;  G[7] = 0.0f;  G[8] = gc[5];
        }
        float ddepth = 0.0f;
        for (int k = 0; k < 3; ++k) {
            const float sc = r.base[k] * dm;
            const float a0 = r.A[k], a1 = r.A[3 + k], a2 = r.A[6 + k];   // column k of A
            // G a_k, G^T a_k
            const float ga0 = G[0] * a0 + G[1] * a1 + G[2] * a2;
            const float ga1 = G[3] * a0 + G[4] * a1 + G[5] * a2;
            const float ga2 = G[6] * a0 + G[7] * a1 + G[8] * a2;
            const float gt0 = G[0] * a0 + G[3] * a1 + G[6] * a2;
            const float gt1 = G[1] * a0 + G[4] * a1 + G[7] * a2;
            const float gt2 = G[2] * a0 + G[5] * a1 + G[8] * a2;
            const float dD = a0 * ga0 + a1 * ga1 + a2 * ga2;
            float ds = 2.0f * sc * dD;
            if (g.scales) ds += g.scales[3 * e + k];
            const float Dk = sc * sc;
            dA[k] += (ga0 + gt0) * Dk;
            dA[3 + k] += (ga1 + gt1) * Dk;
            dA[6 + k] += (ga2 + gt2) * Dk;
            ddepth += ds * r.base[k] * cam.mult;
            dbase[k] += ds * dm;
        }
        const float gm0 = g.means[3 * e], gm1 = g.means[3 * e + 1], gm2 = g.means[3 * e + 2];
        ddepth += gm0 * r.dw[0] + gm1 * r.dw[1] + gm2 * r.dw[2];
        ddw[0] += gm0 * depth; ddw[1] += gm1 * depth; ddw[2] += gm2 * depth;
        o.depths[e] = ddepth;
    }

    float *graw = o.raw + 7 * row;
    for (int k = 0; k < 3; ++k) graw[k] = dbase[k] * (d.scale_max - d.scale_min) * r.sig[k] * (1.0f - r.sig[k]);

    // dR = C^T dA, then R(q) = I + t P(q), t = 2/(q.q + eps)
    float dR[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            dR[3 * a + b] = cam.C[a] * dA[b] + cam.C[3 + a] * dA[3 + b] + cam.C[6 + a] * dA[6 + b];
    const float i = r.q[0], j = r.q[1], k = r.q[2], w = r.q[3], t = r.t;
    const float P[9] = {-(j * j + k * k), i * j - k * w, i * k + j * w,
                        i * j + k * w, -(i * i + k * k), j * k - i * w,
                        i * k - j * w, j * k + i * w, -(i * i + j * j)};
    float dRP = 0.0f;
    for (int m = 0; m < 9; ++m) dRP += dR[m] * P[m];
    const float s01 = dR[1] + dR[3], s02 = dR[2] + dR[6], s12 = dR[5] + dR[7];
    const float a21 = dR[7] - dR[5], a02 = dR[2] - dR[6], a10 = dR[3] - dR[1];
    const float coef = 2.0f * t / r.n * dRP;
    float dq[4];
    dq[0] = t * (j * s01 + k * s02 - 2.0f * i * (dR[4] + dR[8]) + w * a21) - coef * i;
    dq[1] = t * (i * s01 + k * s12 - 2.0f * j * (dR[0] + dR[8]) + w * a02) - coef * j;
    dq[2] = t * (i * s02 + j * s12 - 2.0f * k * (dR[0] + dR[4]) + w * a10) - coef * k;
    dq[3] = t * (k * a10 + j * a02 + i * a21) - coef * w;
    if (g.rotations)
        for (int m = 0; m < 4; ++m) dq[m] += g.rotations[4 * row + m];
    // q = raw * u, u = 1/(len + eps): d raw = u dq - (dq . raw) u^2 raw / len
    const float *rq = in.raw + row * (size_t)d.raw_stride + 3;
    const float dot = dq[0] * rq[0] + dq[1] * rq[1] + dq[2] * rq[2] + dq[3] * rq[3];
    const float back = r.len > 0.0f ? dot * r.u * r.u / r.len : 0.0f;
    for (int m = 0; m < 4; ++m) graw[3 + m] = r.u * dq[m] - back * rq[m];

    // world direction -> camera direction -> normalise -> K^-1 [x y 1]
    float ddc[3];
    for (int a = 0; a < 3; ++a) ddc[a] = cam.C[a] * ddw[0] + cam.C[3 + a] * ddw[1] + cam.C[6 + a] * ddw[2];
    const float proj = ddc[0] * r.dc[0] + ddc[1] * r.dc[1] + ddc[2] * r.dc[2];
    const float iv = 1.0f / r.vlen;
    const float dv0 = (ddc[0] - proj * r.dc[0]) * iv, dv1 = (ddc[1] - proj * r.dc[1]) * iv, dv2 = (ddc[2] - proj * r.dc[2]) * iv;
    o.coordinates[2 * row] = cam.Ki[0] * dv0 + cam.Ki[3] * dv1 + cam.Ki[6] * dv2;
    o.coordinates[2 * row + 1] = cam.Ki[1] * dv0 + cam.Ki[4] * dv1 + cam.Ki[7] * dv2;
}

}  // namespace lsr

using namespace lsr;

static int adapter_check(const lsr_adapter_dims *d, const lsr_adapter_inputs *in) {
    if (!d || !in) return LSR_ENULL;
    if (d->num_cameras < 1 || d->rays < 0 || d->samples < 1 || d->height < 1 || d->width < 1) return LSR_EINVAL;
    if (d->cov_elems != 6 && d->cov_elems != 9) return LSR_EINVAL;
    if (d->raw_stride < 7) return LSR_EINVAL;
    if (d->num_cameras > 65535) return LSR_EUNSUPPORTED;
    if (!in->extrinsics || !in->intrinsics) return LSR_ENULL;
    if (d->rays > 0 && (!in->coordinates || !in->depths || !in->raw)) return LSR_ENULL;
    return LSR_OK;
}

extern "C" {

int lsr_adapter_forward(const lsr_adapter_dims *d, const lsr_adapter_inputs *in,
                        const lsr_adapter_outputs *out, lsr_stream_t stream) {
    note_hip_error(0);
    int rc = adapter_check(d, in);
    if (rc) return rc;
    if (!out) return LSR_ENULL;
    if (d->rays == 0) return LSR_OK;
    if (!out->means || !out->covariances || !out->scales || !out->rotations) return LSR_ENULL;
    const dim3 grid((d->rays + kAdapterThreads - 1) / kAdapterThreads, d->num_cameras);
    hipStream_t s = (hipStream_t)stream;
    prof_begin(kStAdapterFwd, s);
    if (d->cov_elems == 9) hipLaunchKernelGGL(k_adapter_fwd<9>, grid, dim3(kAdapterThreads), 0, s, *d, *in, *out);
    else hipLaunchKernelGGL(k_adapter_fwd<6>, grid, dim3(kAdapterThreads), 0, s, *d, *in, *out);
    prof_end(kStAdapterFwd, s);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { note_hip_error((int)e); return LSR_ELAUNCH; }
    return LSR_OK;
}

int lsr_adapter_backward(const lsr_adapter_dims *d, const lsr_adapter_inputs *in,
                         const lsr_adapter_out_grads *dout, const lsr_adapter_in_grads *din,
                         lsr_stream_t stream) {
    note_hip_error(0);
    int rc = adapter_check(d, in);
    if (rc) return rc;
    if (!dout || !din) return LSR_ENULL;
    if (d->rays == 0) return LSR_OK;
    if (!dout->means || !dout->covariances || !din->coordinates || !din->depths || !din->raw) return LSR_ENULL;
    const dim3 grid((d->rays + kAdapterThreads - 1) / kAdapterThreads, d->num_cameras);
    hipStream_t s = (hipStream_t)stream;
    prof_begin(kStAdapterBwd, s);
    if (d->cov_elems == 9) hipLaunchKernelGGL(k_adapter_bwd<9>, grid, dim3(kAdapterThreads), 0, s, *d, *in, *dout, *din);
    else hipLaunchKernelGGL(k_adapter_bwd<6>, grid, dim3(kAdapterThreads), 0, s, *d, *in, *dout, *din);
    prof_end(kStAdapterBwd, s);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { note_hip_error((int)e); return LSR_ELAUNCH; }
    return LSR_OK;
}

}  // extern "C"
