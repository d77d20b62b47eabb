/*
 * lsr_ply.h — C ABI of the 3DGS `.ply` export (the on-disk format next
 * to the path).  Same library and conventions as lsr_rasterizer.h.
 *
 * Replaces /root/reference/src/model/ply_export.py:26-92 (`export_ply`): per-Gaussian recentring
 * and rescaling (:35-41), the viewer rotation `Rz(-45 deg) @ [[0,0,1],[-1,0,0],[0,-1,0]] @
 * extrinsics[:3,:3]^-1` applied to positions (:43-66) and to the orientation quaternions through
 * rotation matrices (:68-73, scipy `Rotation.from_quat / from_matrix / as_quat`, output order
 * w,x,y,z), DC band of the harmonics (:77), log-scales (:87), and the 17-float vertex record of
 * `construct_list_of_attributes(0)` (:13-23): x y z nx ny nz f_dc_0..2 opacity scale_0..2 rot_0..3.
 * The reference does this with torch + scipy + a Python tuple list per Gaussian and writes through
 * `plyfile`; here one kernel packs the vertex records on the device and a host function writes the
 * binary little-endian file.  The two global statistics (median, 0.95-quantile) are the caller's
 * (two torch reductions) and are passed in as device scalars.
 */
#ifndef LSR_PLY_H
#define LSR_PLY_H

#include "lsr_rasterizer.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LSR_PLY_VERTEX_FLOATS 17

typedef struct lsr_ply_inputs {
    const float *extrinsics;  /* [4][4] camera-to-world of the reference view */
    const float *means;       /* [n][3] */
    const float *scales;      /* [n][3] */
    const float *rotations;   /* [n][4] xyzw */
    const float *harmonics;   /* [n][3][d_sh]; only coefficient 0 of each channel is exported */
    const float *opacities;   /* [n] */
    const float *center;      /* [3]  device: means.median(dim=0) */
    const float *scale_factor;/* [1]  device: means(centred).abs().quantile(0.95, dim=0).max() */
} lsr_ply_inputs;

/* Fill vertices[n][LSR_PLY_VERTEX_FLOATS] on the device.  Asynchronous. */
int lsr_ply_pack(int64_t n, int32_t d_sh, const lsr_ply_inputs *in, float *vertices, lsr_stream_t stream);

/* Write a binary little-endian PLY with one `vertex` element of n records from HOST memory
 * (header as plyfile emits it for float32 properties).  Returns LSR_EINVAL if the file cannot be
 * created or written. */
int lsr_ply_write_host(const char *path, const float *vertices_host, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* LSR_PLY_H */
