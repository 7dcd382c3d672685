/*
 * okvis_amd_frontend.h — C-ABI of the batched reprojection pieces of the OKVIS frontend.
 *
 * The frontend calls the backend's reprojection algebra once per candidate match, one at a time, from the matcher's inner
 * loop.  These entries do the same arithmetic for ALL candidates of one (frame A, frame B, camera pair) in one launch:
 *
 *   okvis_fe_stereo_triangulate  ProbabilisticStereoTriangulator<G>::resetFrames / stereoTriangulate / getUncertainty
 *                                (okvis_frontend/src/ProbabilisticStereoTriangulator.cpp:127-170, 178-250, 253-355, 358-385;
 *                                 triangulateFast: okvis_frontend/src/stereo_triangulation.cpp:50-137)
 *   okvis_fe_project_landmarks   VioKeyframeWindowMatchingAlgorithm<G>::doSetup, Match3D2D branch
 *                                (okvis_frontend/src/VioKeyframeWindowMatchingAlgorithm.cpp:165-213)
 *   okvis_fe_gate_3d2d           ... ::verifyMatch (:320-337) and the gate of ::setBestMatch (:494-512)
 *
 * What stays with the caller is what needs the estimator's book-keeping or image data: which keypoints carry a landmark,
 * descriptor distances, the best-match search, addLandmark / addObservation.  G = PinholeCamera<D> with the distortion
 * models of okvis_amd_ba.h.  All arithmetic is IEEE double like the reference; keypoints are float like cv::KeyPoint.
 * Plain pointers and sizes, int status codes (okvis_amd_ba.h), host buffers in and out; no CPU path.
 */
#ifndef OKVIS_AMD_FRONTEND_H_
#define OKVIS_AMD_FRONTEND_H_

#include <stdint.h>

#include "okvis_amd_ba.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct okvis_fe_context okvis_fe_context; /* one HIP stream + device scratch that grows on demand */

/* PinholeCamera<D> (okvis_cv/include/okvis/cameras/PinholeCamera.hpp): what project / backProject read */
typedef struct okvis_fe_camera {
  double intr[12]; /* fu fv cu cv d0..d7 */
  int32_t model;   /* OKVIS_BA_DIST_* */
  int32_t width;   /* image size: CameraBase::isInImage (implementation/CameraBase.hpp:95-104) */
  int32_t height;
  int32_t reserved;
} okvis_fe_camera;

/* flags of okvis_fe_stereo_triangulate, one byte per candidate */
#define OKVIS_FE_TRI_VALID 1u           /* stereoTriangulate returned true (:178-236)                                   */
#define OKVIS_FE_TRI_NOT_PARALLEL 2u    /* outCanBeInitializedInaccuarate of the 5-argument overload = !isParallel (:208) */
#define OKVIS_FE_TRI_CAN_INIT 4u        /* outCanBeInitialized of the 6-argument overload (:239-250): getUncertainty's
                                           decision AND not parallel                                                    */
#define OKVIS_FE_TRI_RANK_DEFICIENT 8u  /* H.colPivHouseholderQr().rank() < 9 (:349-352): cov is not written            */

/* ProjectionStatus (okvis_cv/include/okvis/cameras/CameraBase.hpp:67-74) as okvis_fe_project_landmarks reports it */
#define OKVIS_FE_PROJ_SUCCESSFUL 0
#define OKVIS_FE_PROJ_OUTSIDE_IMAGE 1
#define OKVIS_FE_PROJ_MASKED 2 /* never produced: masks are not part of this interface */
#define OKVIS_FE_PROJ_BEHIND 3
#define OKVIS_FE_PROJ_INVALID 4

/* flags of okvis_fe_gate_3d2d */
#define OKVIS_FE_GATE_VERIFIED 1u  /* verifyMatch: (int)chi2 < 4 (:333-337)                                  */
#define OKVIS_FE_GATE_ACCEPTED 2u  /* setBestMatch does not return at chi2 > 4.0 (:505-508)                  */
#define OKVIS_FE_GATE_UNCERTAIN 4u /* U_tot.norm() > 25 / (sigma_B^2 sqrt 2): numUncertainMatches_++ (:511) */

int okvis_fe_create(okvis_fe_context** out, int device);
void okvis_fe_destroy(okvis_fe_context* ctx);

/* For every candidate (a, b) = pairs[i]: keypoint a of image A against keypoint b of image B.
 *   T_AB[7]     r, q(xyzw) of T_CaCb (resetFrames :127-136)
 *   UOplus[36]  6x6 covariance of T_AB in its tangent space (row-major, symmetric positive definite); the reference turns
 *               it into the Gauss-Newton block H_(0:6,0:6) through a PoseError linearised at T_AB (:142-160)
 *   kp_a/kp_b   [n][3] float: x, y, size of the cv::KeyPoints (sigma = 0.8 size / 12)
 *   sigma_ray   [n_pairs] or NULL; NULL or -1.0 = the triangulator's own 0.5 / min(fu_A, fu_B) (:162-166, :187-190)
 * Outputs (any may be NULL): hp_a [n_pairs][4] homogeneous point in A (written when VALID or when the reprojection check
 * is what failed, like the reference's out parameter), cov [n_pairs][9] row-major 3x3 UOplus of the point (written when
 * VALID and not RANK_DEFICIENT), flags [n_pairs].  want_uncertainty = 0 stops after stereoTriangulate (verifyMatch, :310-317). */
int okvis_fe_stereo_triangulate(okvis_fe_context* ctx, const okvis_fe_camera* cam_a, const okvis_fe_camera* cam_b,
                                const double* T_AB, const double* UOplus, int32_t n_a, const float* kp_a, int32_t n_b,
                                const float* kp_b, int32_t n_pairs, const int32_t* pairs, const double* sigma_ray,
                                int32_t want_uncertainty, double* hp_a, double* cov, uint8_t* flags);
/* The same call with one more output: gn [n_pairs][81] = the 9x9 Gauss-Newton matrix H of getUncertainty (:286-345; rows /
 * columns 0..5 the relative pose, 6..8 the point), row-major, zeros where getUncertainty does not run.  For referees of the
 * point covariance (tests/test_gpu_frontend.py inverts it in extended precision); NULL = okvis_fe_stereo_triangulate. */
int okvis_fe_stereo_triangulate_gn(okvis_fe_context* ctx, const okvis_fe_camera* cam_a, const okvis_fe_camera* cam_b,
                                const double* T_AB, const double* UOplus, int32_t n_a, const float* kp_a, int32_t n_b,
                                const float* kp_b, int32_t n_pairs, const int32_t* pairs, const double* sigma_ray,
                                int32_t want_uncertainty, double* hp_a, double* cov, uint8_t* flags, double* gn);

/* hp_W [n][4] through T_CbW [7] into camera B: uv [n][2] (written unless INVALID), U [n][4] = J P_C J^T row-major 2x2 with
 * P_C = P3 [9] (row-major 3x3, the top-left block of the relative pose uncertainty, :197-203), status [n]. */
int okvis_fe_project_landmarks(okvis_fe_context* ctx, const okvis_fe_camera* cam_b, const double* T_CbW, const double* P3,
                               int32_t n, const double* hp_W, double* uv, double* U, uint8_t* status);

/* candidates (a, b) = pairs[i]: projection a (uv, U from okvis_fe_project_landmarks) against keypoint b of image B.
 * chi2 [n_pairs] = err^T (sigma_B^2 I + U_a)^-1 err, flags [n_pairs]. */
int okvis_fe_gate_3d2d(okvis_fe_context* ctx, int32_t n_proj, const double* uv, const double* U, int32_t n_b,
                       const float* kp_b, int32_t n_pairs, const int32_t* pairs, double* chi2, uint8_t* flags);

#ifdef __cplusplus
}
#endif
#endif /* OKVIS_AMD_FRONTEND_H_ */
