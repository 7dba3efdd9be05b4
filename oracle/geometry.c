/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the map-fusion hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Plain-C restatement of the per-pixel geometry of the reference's
 * SemanticMapper.update_map (sem_objnav/obj_nav/mapper.py):
 *   - depth sentinel zeroing            mapper.py:282-283
 *   - pointcloud()                      mapper.py:18-32
 *   - rotate + translate                mapper.py:305-311
 *   - world2map()                       mapper.py:851-863
 *   - height cut-off / bounds masks     mapper.py:319-329
 *
 * All arithmetic is IEEE binary64 with contraction disabled (build with
 * -ffp-contract=off).  The 3x3 rotation is the k-ordered fused chain
 *   out_k = fma(m[k][2], z, fma(m[k][1], y, m[k][0] * x))
 * which is what numpy's tensordot (OpenBLAS dgemm, the path numpy-quaternion's
 * rotate_vectors takes) produces bit for bit; tests/golden/make_golden.py checks
 * that equality when it records the vectors.
 */
#include <math.h>
#include <stdint.h>

/* One frame.  depth is modified in place exactly like the reference does.
 * rot: row-major 3x3, trans: camera position, lower: map_bounds[:3].
 * Outputs per pixel (flat index i = r*W + c):
 *   row[i], col[i]  int64 cell indices (row <-> world z, col <-> world x)
 *   height[i]       world y (f64)
 *   valid[i]        depth > 0 after sentinel zeroing
 */
void oracle_project(float *depth, int32_t H, int32_t W,
                    float depth_min, float depth_max,
                    double fx, double fy,
                    const double *rot, const double *trans, const double *lower,
                    double pixel2meter,
                    int64_t *row, int64_t *col, double *height, uint8_t *valid)
{
    const double half_w = (double)W / 2.0;
    const double half_h = (double)H / 2.0;
    for (int32_t r = 0; r < H; ++r) {
        for (int32_t c = 0; c < W; ++c) {
            const int64_t i = (int64_t)r * W + c;
            float d = depth[i];
            if (d == depth_min) d = 0.0f;
            if (d == depth_max) d = 0.0f;
            depth[i] = d;
            const int ok = d > 0.0f;
            double x = 0.0, y = 0.0, z = 0.0;
            if (ok) {
                z = (double)d;
                x = z * ((double)c - half_w) / fx;
                y = z * ((double)r - half_h) / fy;
            }
            const double wx = fma(rot[2], z, fma(rot[1], y, rot[0] * x)) + trans[0];
            const double wy = fma(rot[5], z, fma(rot[4], y, rot[3] * x)) + trans[1];
            const double wz = fma(rot[8], z, fma(rot[7], y, rot[6] * x)) + trans[2];
            /* numpy's astype(int) on x86-64 (cvttsd2si): NaN, infinities and values beyond the
             * int64 range become INT64_MIN; this cast compiles to the same instruction */
            row[i] = (int64_t)floor((wz - lower[2]) / pixel2meter);
            col[i] = (int64_t)floor((wx - lower[0]) / pixel2meter);
            height[i] = wy;
            valid[i] = (uint8_t)ok;
        }
    }
}
