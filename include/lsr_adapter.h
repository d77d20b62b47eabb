/*
 * lsr_adapter.h — C ABI of the "Gaussian adapter tail": the per-Gaussian geometry that produces
 * exactly the rasterizer's inputs.  Same library (liblsr_hip.so), same
 * conventions as lsr_rasterizer.h: device pointers, sizes, a hipStream_t, negative LSR_E* codes.
 *
 * What it replaces in the reference (paths relative to /root/reference), all of it a chain of
 * ~45 elementwise / tiny-matmul PyTorch ops over (b, v, r, srf, spp) tensors in forward and as
 * many again in autograd's backward:
 *   - src/model/encoder/common/gaussian_adapter.py:78-85   scale range map
 *         scales = (min + (max-min) * sigmoid(raw)) * depth * get_scale_multiplier(K, pixel_size)
 *   - gaussian_adapter.py:116-127  get_scale_multiplier = 0.1 * sum(K[:2,:2]^-1 @ (1/w, 1/h))
 *   - gaussian_adapter.py:88       rotations = raw / (|raw| + eps)
 *   - src/model/encoder/common/gaussians.py:8-31   quaternion_to_matrix (xyzw order,
 *         two_s = 2 / (q.q + eps))
 *   - gaussians.py:34-44           build_covariance = R S S^T R^T
 *   - gaussian_adapter.py:96-98    covariances = c2w_rot @ cov @ c2w_rot^T
 *   - gaussian_adapter.py:101-102 + src/geometry/projection.py:74-114 (unproject, get_world_rays)
 *         means = origin + normalize(K^-1 [x y 1]) rotated to world * depth
 *   - src/model/decoder/cuda_splatting.py:148,157   the per-view `triu_indices` gather of the
 *         covariance: cov_elems = 6 emits the packed upper triangle (xx,xy,xz,yy,yz,zz) directly.
 * Not covered (stays in PyTorch): the SH coefficient masks and `rotate_sh` (e3nn Wigner-D,
 * gaussian_adapter.py:92-94,107-108) — e3nn is not a dependency of this library.
 *
 * Element indexing.  `num_cameras` context cameras (b*v); per camera `rays` parameter rows
 * (r*srf); per row `samples` depth samples (spp) that share the row's raw scale / rotation /
 * coordinate (the reference broadcasts them over spp, encoder_epipolar.py:186-193).
 * e = (camera * rays + ray) * samples + sample.
 */
#ifndef LSR_ADAPTER_H
#define LSR_ADAPTER_H

#include "lsr_rasterizer.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lsr_adapter_dims {
    int32_t num_cameras;    /* >= 1 */
    int32_t rays;           /* rows per camera, >= 0 */
    int32_t samples;        /* depth samples per row, >= 1 */
    int32_t height, width;  /* `image_shape` -> pixel_size = (1/w, 1/h) (gaussian_adapter.py:82-83) */
    int32_t cov_elems;      /* 9: (...,3,3) as the reference's Gaussians.covariances; 6: packed triu */
    float scale_min, scale_max; /* cfg.gaussian_scale_min / max */
    float eps;              /* forward()'s eps (1e-8) for the quaternion normalisation */
    int32_t raw_stride;     /* floats between consecutive rows of `raw` (>= 7; the reference passes a
                             * strided view `gaussians[..., 2:]` of the Linear output) */
    int32_t reserved0, reserved1;
} lsr_adapter_dims;

typedef struct lsr_adapter_inputs {
    const float *extrinsics;   /* [cam][4][4] camera-to-world */
    const float *intrinsics;   /* [cam][3][3] normalised */
    const float *coordinates;  /* [cam][rays][2]  xy in [0,1] */
    const float *depths;       /* [cam][rays][samples] */
    const float *raw;          /* [cam][rays][raw_stride]: +0..2 raw scales, +3..6 raw quaternion xyzw */
} lsr_adapter_inputs;

typedef struct lsr_adapter_outputs {
    float *means;        /* [cam][rays][samples][3] */
    float *covariances;  /* [cam][rays][samples][cov_elems] */
    float *scales;       /* [cam][rays][samples][3]   (Gaussians.scales) */
    float *rotations;    /* [cam][rays][4]            (Gaussians.rotations before its broadcast) */
} lsr_adapter_outputs;

/* Upstream gradients; means/covariances required, scales/rotations may be NULL (= zero). */
typedef struct lsr_adapter_out_grads {
    const float *means, *covariances, *scales, *rotations;
} lsr_adapter_out_grads;

/* Input gradients, all written (not accumulated); sums over `samples` happen in-kernel. */
typedef struct lsr_adapter_in_grads {
    float *coordinates;  /* [cam][rays][2] */
    float *depths;       /* [cam][rays][samples] */
    float *raw;          /* [cam][rays][7] dense: scales 0..2, quaternion 3..6 */
} lsr_adapter_in_grads;

/* Forward: one launch, asynchronous on `stream`. */
int lsr_adapter_forward(const lsr_adapter_dims *d, const lsr_adapter_inputs *in,
                        const lsr_adapter_outputs *out, lsr_stream_t stream);

/* Backward of the above w.r.t. coordinates, depths and the 7 raw parameters (cameras are data). */
int lsr_adapter_backward(const lsr_adapter_dims *d, const lsr_adapter_inputs *in,
                         const lsr_adapter_out_grads *dout, const lsr_adapter_in_grads *din,
                         lsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LSR_ADAPTER_H */
