/*
 * okvis_amd_ba.h — C-ABI of the MI355X-native sliding-window bundle-adjustment backend.
 *
 * This is the drop-in boundary for ONE path of ethz-asl/okvis: the optimisation loop behind
 * okvis::Estimator::optimize (reference okvis_ceres/src/Estimator.cpp:843-906 -> Map::solve,
 * okvis_ceres/include/okvis/ceres/Map.hpp:371-373 -> ::ceres::Solve).  The reference has no FFI; the
 * boundary a maintainer binds is the concrete C++ class okvis::Estimator
 * (okvis_ceres/include/okvis/Estimator.hpp:77-581).  The replacement Estimator (okvis_amd/csrc/host/)
 * keeps that class's methods and calls ONLY the functions declared here.  Plain pointers and sizes,
 * no C++/torch types, int status codes, no exceptions across the boundary.
 *
 * All floating point data are IEEE double, like the reference.  Indices are int32, times are int64
 * nanoseconds (okvis::Time is {u32 sec, u32 nsec}, okvis_time/include/okvis/Time.hpp:125-205).
 *
 * Block conventions (reference file:line):
 *   pose block   double[7] = r_x r_y r_z q_x q_y q_z q_w   (PoseParameterBlock.cpp:68-79); minimal dim 6,
 *                (+) = PoseLocalParameterization::plus (PoseLocalParameterization.cpp:60-87).  Both the
 *                body poses T_WS and the camera extrinsics T_SC are pose blocks and live in ONE array.
 *   speed/bias   double[9] = v_W b_g b_a                   (SpeedAndBiasParameterBlock.hpp:108-153), Euclidean.
 *   landmark     double[4] homogeneous x y z w in W        (HomogeneousPointLocalParameterization.cpp:59-71);
 *                minimal dim 3, w is never updated.
 */
#ifndef OKVIS_AMD_BA_H_
#define OKVIS_AMD_BA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OKVIS_BA_ABI_VERSION 7

/* status codes (0 ok; >0 = hipError_t passthrough + 1000; <0 = argument / state errors) */
#define OKVIS_BA_OK 0
#define OKVIS_BA_ERR_ARG (-1)          /* null pointer / negative size / index out of range            */
#define OKVIS_BA_ERR_STATE (-2)        /* call order violated (e.g. optimize before upload)             */
#define OKVIS_BA_ERR_UNSUPPORTED (-3)  /* structure exceeds a documented limit (see okvis_ba_limits)    */
#define OKVIS_BA_ERR_NO_DEVICE (-4)    /* no HIP device: the product path has NO CPU fallback           */
#define OKVIS_BA_ERR_NUMERIC (-5)      /* reduced system not positive definite after max damping        */
#define OKVIS_BA_HIP_ERROR_BASE 1000

/* camera distortion models (okvis_cv/include/okvis/cameras/ XxxDistortion.hpp) */
#define OKVIS_BA_DIST_NONE 0
#define OKVIS_BA_DIST_RADTAN 1       /* RadialTangentialDistortion  k1 k2 p1 p2            */
#define OKVIS_BA_DIST_EQUIDISTANT 2  /* EquidistantDistortion       k1 k2 k3 k4            */
#define OKVIS_BA_DIST_RADTAN8 3      /* RadialTangentialDistortion8 k1 k2 p1 p2 k3 k4 k5 k6 */

/* block types inside the marginalisation prior */
/* one ImuError preintegration as the device keeps it (Delta_q, the integrals, their bias Jacobians, the square-root information,
 * the reference bias), opaque to the caller: okvis_ba_fetch_imu_caches hands it out, okvis_ba_window::imu_cache takes it back */
#define OKVIS_BA_IMU_CACHE_DOUBLES 290

#define OKVIS_BA_BLOCK_POSE 0
#define OKVIS_BA_BLOCK_SPEEDBIAS 1

/* IMU noise parameters that reach the hot path (okvis_common/include/okvis/Parameters.hpp ImuParameters;
 * used at ImuError.cpp:153-173,228-249,564) */
typedef struct okvis_ba_imu_params {
  double sigma_g_c;  /* gyro noise density                */
  double sigma_a_c;  /* accelerometer noise density       */
  double sigma_gw_c; /* gyro drift noise density          */
  double sigma_aw_c; /* accelerometer drift noise density */
  double g;          /* gravity magnitude                 */
  double g_max;      /* gyro saturation                   */
  double a_max;      /* accelerometer saturation          */
} okvis_ba_imu_params;

/*
 * One sliding window as flat struct-of-arrays.  This replaces okvis::ceres::Map's hash maps of
 * shared_ptr parameter blocks and residual blocks (Map.hpp:348-402, Map.cpp:292-565) and
 * Estimator::statesMap_/landmarksMap_ (Estimator.hpp:555-563).  All pointers are HOST pointers owned by
 * the caller; okvis_ba_upload copies them.
 */
typedef struct okvis_ba_window {
  /* ---- parameter blocks ---- */
  int32_t n_pose;             /* pose-type blocks (T_WS per frame + T_SC extrinsics)                    */
  const double* pose;         /* [n_pose][7]                                                           */
  const uint8_t* pose_fixed;  /* [n_pose] 1 = Map::setParameterBlockConstant (Map.cpp:568-576)         */
  int32_t n_sb;
  const double* sb;           /* [n_sb][9]                                                             */
  const uint8_t* sb_fixed;    /* [n_sb]                                                                */
  int32_t n_lm;
  const double* lm;           /* [n_lm][4]                                                             */

  /* ---- cameras (PinholeCamera<D>, okvis_cv/.../implementation/PinholeCamera.hpp:148-226) ---- */
  int32_t n_cam;
  const double* cam_intr;     /* [n_cam][12] = fu fv cu cv d0..d7                                      */
  const int32_t* cam_model;   /* [n_cam] OKVIS_BA_DIST_*                                               */

  /* ---- reprojection observations (ReprojectionError<G>, implementation/Estimator.hpp:43-90) ----
   * MUST be sorted by (lm, pose, cam) (repeated triples allowed: two keypoints of one image matched to one landmark);
   * every landmark referenced by <= okvis_ba_limits.max_obs_per_lm. */
  int32_t n_obs;
  const int32_t* obs_lm;      /* [n_obs] landmark index                                                */
  const int32_t* obs_pose;    /* [n_obs] pose block index of T_WS                                      */
  const int32_t* obs_ext;     /* [n_obs] pose block index of T_SC                                      */
  const int32_t* obs_cam;     /* [n_obs] camera (intrinsics) index                                     */
  const double* obs_uv;       /* [n_obs][2] keypoint measurement                                       */
  const double* obs_sqrtw;    /* [n_obs] sqrt-information scalar = 8/keypoint.size
                                 (information = 64/size^2 * I2, implementation/Estimator.hpp:62-65)     */
  double cauchy_b;            /* CauchyLoss(b) on reprojection residuals (Estimator.cpp:60: 1.0);
                                 <= 0 disables the loss                                                 */

  /* ---- IMU factors (ImuError over pose0,sb0,pose1,sb1; Estimator.cpp:288-307) ---- */
  int32_t n_imu;
  const int32_t* imu_pose0;   /* [n_imu] */
  const int32_t* imu_sb0;
  const int32_t* imu_pose1;
  const int32_t* imu_sb1;
  const int64_t* imu_t0;      /* [n_imu] ns */
  const int64_t* imu_t1;
  const int32_t* imu_s_begin; /* [n_imu] first raw sample of this factor's measurement deque           */
  const int32_t* imu_s_count; /* [n_imu] number of raw samples (the deque copied at ImuError.hpp:151)   */
  int32_t n_imu_samples;
  const int64_t* imu_s_t;     /* [n_imu_samples] ns */
  const double* imu_s_gyr;    /* [n_imu_samples][3] */
  const double* imu_s_acc;    /* [n_imu_samples][3] */
  okvis_ba_imu_params imu_params;

  /* ---- absolute pose priors (PoseError, PoseError.cpp:86-138; Estimator.cpp:238-262) ---- */
  int32_t n_pprior;
  const int32_t* pprior_pose;      /* [n_pprior] pose block index                                      */
  const double* pprior_meas;       /* [n_pprior][7]                                                    */
  const double* pprior_sqrtinfo;   /* [n_pprior][36] row-major upper-triangular L^T (PoseError.cpp:70-76)*/

  /* ---- speed/bias priors (SpeedAndBiasError.cpp:89-118; Estimator.cpp:269-284) ---- */
  int32_t n_sbprior;
  const int32_t* sbprior_sb;
  const double* sbprior_meas;      /* [n_sbprior][9]  */
  const double* sbprior_sqrtinfo;  /* [n_sbprior][81] */

  /* ---- relative pose factors (RelativePoseError.cpp:84-163; Estimator.cpp:310-336) ---- */
  int32_t n_relpose;
  const int32_t* rel_pose0;
  const int32_t* rel_pose1;
  const double* rel_sqrtinfo;      /* [n_relpose][36] */

  /* ---- marginalisation prior (MarginalizationError::EvaluateWithMinimalJacobians,
   *      MarginalizationError.cpp:893-946): e = e0 + J * DeltaChi ---- */
  int32_t marg_dim;                /* D_m = rows = cols of J (0 = no prior)                            */
  int32_t marg_nblocks;
  const int32_t* marg_block_type;  /* [marg_nblocks] OKVIS_BA_BLOCK_*                                  */
  const int32_t* marg_block_idx;   /* [marg_nblocks] index into pose[] / sb[]                          */
  const int32_t* marg_block_off;   /* [marg_nblocks] column offset (orderingIdx) of the block in J     */
  const double* marg_J;            /* [marg_dim][marg_dim] row-major                                   */
  const double* marg_e0;           /* [marg_dim]                                                       */
  const double* marg_lin;          /* [marg_nblocks][9] linearisation points (7 or 9 used)             */

  /* ImuError keeps a mutable preintegration cache whose reference bias (speedAndBiases_ref_, ImuError.hpp) is
   * the bias of its last redoPreintegration and survives optimize() calls.  Optional: the reference bias each
   * factor's cache starts from (download it after optimize with OKVIS_BA_ARR_IMU_SB_REF).  NULL / flag 0 = a
   * brand-new factor: the first evaluation re-preintegrates at the current bias (redo_ = true, ImuError.cpp:62).
   * Flag 1 = the reference bias only: the cache is rebuilt AT that reference on first use (what the reference's object would
   * still hold).  Flag 2 = the preintegration itself travels too — the record okvis_ba_fetch_imu_caches handed out for the same
   * term (same samples, same interval) sits in imu_cache: nothing is rebuilt unless the bias has moved past the threshold
   * (ImuError.cpp:549), exactly like the reference's ImuError object that lives on between optimize() calls.  The values are
   * the same either way (the rebuild is deterministic); flag 2 saves the 0.1 ms a re-preintegration takes. */
  const double* imu_sb_ref;        /* [n_imu][9] or NULL */
  const uint8_t* imu_sb_ref_valid; /* [n_imu] or NULL: 0 / 1 / 2, see above */
  const double* imu_cache;         /* [n_imu][OKVIS_BA_IMU_CACHE_DOUBLES] or NULL (opaque records, read where the flag is 2) */
} okvis_ba_window;

/*
 * Solver policy.  The reference configures Ceres 1.9 TRUST_REGION / DOGLEG / SPARSE_SCHUR with defaults
 * (Estimator.cpp:854-873: trust_region_strategy_type = DOGLEG, dogleg_type TRADITIONAL_DOGLEG, jacobi_scaling on);
 * Ceres is not in the reference tree.  The default here is that policy, restated from Ceres' documentation / public
 * sources (DESIGN.md "solver policy" lists what is unverified): traditional dogleg between the Cauchy point and the
 * Gauss-Newton point of the Schur-reduced system in Jacobi-scaled, D-normalised variables.  The Levenberg-Marquardt
 * safeguard of round 1 stays available as OKVIS_BA_STRATEGY_LM.
 */
#define OKVIS_BA_STRATEGY_DOGLEG 0
#define OKVIS_BA_STRATEGY_LM 1
/*
 * Launch shapes and A/B switches (ABI 7).  Up to ABI 6 these were some twenty environment variables read inside the library: a
 * linked-in backend must not change its behaviour with its host process's environment, so they are fields now, set per solver
 * with okvis_ba_set_options BEFORE the upload they are meant for (okvis_ba_check_window[_lists] take them through their options
 * argument).  All zero = the defaults.  None of them changes what is computed; a different grouping of a sum moves its rounding.
 * Every field is driven against the oracle by tests/test_gpu_tuning.py.  The library reads ONE environment variable, OKVIS_BA_DEBUG
 * (a comma-separated list of print-only diagnostics: "build", "upload", "marg", "arena=<file>"); it never changes a result.
 */
#define OKVIS_BA_TUNE_SCHUR_DECIDES 0x1u      /* the separate Schur launch takes the trust-region decision itself (rounds 1-4) instead of
                                                 reducing the trial buffer decision-free (DESIGN.md section 5, row 1')                    */
#define OKVIS_BA_TUNE_SCHUR_VALU 0x2u         /* schur_kernel (fp64 FMA) also where the matrix-core kernel would run                      */
#define OKVIS_BA_TUNE_SCHUR_MFMA_LARGE 0x4u   /* the matrix-core Schur kernel also for pose parts of several 96-row tiles (configs[2])    */
#define OKVIS_BA_TUNE_NO_LDL_COMP 0x8u        /* no compensated elimination in the dense solver's prior-carrying diagonal blocks          */
#define OKVIS_BA_TUNE_LDL_COMP_ALL 0x10u      /* compensated elimination in every diagonal block                                          */
#define OKVIS_BA_TUNE_H0_ON_HOST 0x20u        /* H0 = J^T J of a large marginalisation prior on the host instead of marg_h0_kernel        */
#define OKVIS_BA_TUNE_NO_EARLY_PREINTEGRATION 0x40u /* a new IMU term's first preintegration inside the first linearise launch, not at upload */
#define OKVIS_BA_TUNE_NO_MARG_TILES 0x80u     /* okvis_ba_marginalize: kept blocks > 96 rows on the single workgroup, not on the tiled tail */
#define OKVIS_BA_TUNE_NO_SMALL_RIDE 0x100u    /* IMU / prior factors evaluated in small_kernel behind the solve launch (rounds 4-5), not inside
                                                 the decision-free Schur launch (schur_ride_kernel)                                          */
#define OKVIS_BA_SOLVE_AUTO 0                  /* the library picks (OKVIS_BA_ROUTE_SOLVE_MODE reports what it picked)                     */
#define OKVIS_BA_SOLVE_DENSE 1                /* blocked LDL^T of the whole D x D reduced system in LDS (rounds 3-5)                      */
#define OKVIS_BA_SOLVE_CHAIN 2                /* speed/bias blocks eliminated along the IMU chain first, dense pose system behind it      */
typedef struct okvis_ba_tuning {
  uint32_t flags;               /* OKVIS_BA_TUNE_* */
  int32_t fused_max_windows;    /* batches up to this many windows use the fused linearise + reduce launch; 0 = 48, < 0 = never  */
  int32_t group_lm;             /* landmarks per linearise group; 0 = auto (16 for <= 8 windows, else 32), at most 64            */
  int32_t group_work;           /* > 0: close a linearise group at this many observations x blocks (work-balanced groups); 0 = off */
  int32_t split_small_min;      /* batches from this many windows on run the IMU / prior factors in a launch of their own; 0 = 40 */
  int32_t lin2_occupancy;       /* piece-path linearise kernel built for 4 (default, 0) or 3 workgroups per CU                   */
  int32_t stagger_us;           /* start offset between the sub-batch streams; 0 = 10 us, < 0 = none                              */
  int32_t solve_mode;           /* reduced solve: 0 = auto, see OKVIS_BA_SOLVE_*                                                  */
} okvis_ba_tuning;
typedef struct okvis_ba_options {
  double initial_radius;        /* 1e4   */
  double max_radius;            /* 1e16  */
  double min_radius;            /* 1e-32 */
  double min_lm_diagonal;       /* 1e-6  */
  double max_lm_diagonal;       /* 1e32  */
  double min_relative_decrease; /* 1e-3  */
  double function_tolerance;    /* 1e-6  (0 disables: every launched iteration does full work)          */
  double gradient_tolerance;    /* 1e-10 */
  double parameter_tolerance;   /* 1e-8  */
  int32_t use_graph;            /* 1 = replay the captured hipGraph of the iteration sequence           */
  int32_t schur_lm_per_block;   /* landmarks per Schur workgroup (0 = auto)                             */
  int32_t debug_arrays;         /* 1 = also write the parity/debug arrays (per-observation residuals,
                                   damped reduced matrix); 2 = clock64() phase stamps only (diagnostics,
                                   OKVIS_BA_ARR_PROF); 0 in production                                    */
  int32_t gauss_newton;         /* 1 = plain Gauss-Newton: every step is accepted and the damping radius stays
                                   at initial_radius (no trust-region logic); used by bench.py so that
                                   every timed iteration performs identical, full work                   */
  int32_t n_streams;            /* sub-batches of windows on separate HIP streams (phases of different
                                   windows overlap); 0 = auto (3 for >= 56 windows, 2 for >= 16, else 1)  */
  int32_t fp32_linearize;       /* BASELINE configs[4] (mixed-precision study): 1 = reprojection residuals, Jacobians
                                   and their J^T J / J^T r accumulation in fp32; state, Schur complement and the
                                   reduced solve stay fp64.  0 (default) = everything fp64 like the reference   */
  int32_t strategy;             /* OKVIS_BA_STRATEGY_DOGLEG (default, what Estimator.cpp:858 configures) or _LM   */
  int32_t jacobi_scaling;       /* 1 (default, Ceres Solver::Options::jacobi_scaling): columns scaled by
                                   1/(1+sqrt(diag J^T J)) of the FIRST linearisation of the optimize() call;
                                   dogleg strategy only                                                          */
  int32_t max_consecutive_invalid_steps; /* 5 (Ceres max_num_consecutive_invalid_steps): then termination 5     */
  int32_t reserved0;            /* diagnostics.  Bit 2 (value 4): keep the landmark Schur reduction in its own launch
                                   (default: DOGLEG / fixed-radius runs whose windows fit reduce each group inside the
                                   linearise launch).  Bit 3 (8): the staged linearise kernel instead
                                   of the piece path.  Bit 4 (16): solving workgroups do not wait for their helper
                                   workgroups (exercises the time-out route).  Bits 0-1: not read any more           */
  okvis_ba_tuning tuning;       /* launch shapes and A/B switches; all zero = the defaults (see above)              */
} okvis_ba_options;

/* per-window result of okvis_ba_optimize (what ::ceres::Solver::Summary gives Estimator::optimize) */
typedef struct okvis_ba_summary {
  double initial_cost;
  double final_cost;
  int32_t iterations;            /* iterations executed (successful + unsuccessful), excluding iteration 0 */
  int32_t successful_steps;
  int32_t termination;           /* 0 = max iterations, 1 = function tol, 2 = gradient tol, 3 = parameter tol,
                                    4 = radius below min, 5 = numeric failure */
  int32_t reserved;
  double final_radius;
  double gradient_max_norm;
} okvis_ba_summary;

typedef struct okvis_ba_limits {
  int32_t max_obs_per_lm;   /* observations of one landmark                          */
  int32_t max_reduced_dim;  /* 6*free poses + 9*free speed/bias of ONE window        */
  int32_t max_marg_dim;
  int32_t max_imu_samples_per_factor;
} okvis_ba_limits;

typedef struct okvis_ba_solver okvis_ba_solver; /* opaque: a batch of independent windows on one GPU */

/* identify array for okvis_ba_download() — intermediate results used by the parity tests */
enum okvis_ba_array {
  OKVIS_BA_ARR_POSE = 0,        /* [n_pose][7]  accepted state                                          */
  OKVIS_BA_ARR_SB = 1,          /* [n_sb][9]                                                            */
  OKVIS_BA_ARR_LM = 2,          /* [n_lm][4]                                                            */
  OKVIS_BA_ARR_OBS_RESIDUAL = 3,/* [n_obs][2]   un-robustified weighted residual at the accepted state  */
  OKVIS_BA_ARR_LM_V = 4,        /* [n_lm][6]    upper-tri V = sum J_l^T J_l (robustified)               */
  OKVIS_BA_ARR_LM_B = 5,        /* [n_lm][3]    sum J_l^T r                                             */
  OKVIS_BA_ARR_LM_HQ = 6,       /* [n_lm][6]    un-robustified J_l^T J_l (Map::getLhs, Map.cpp:101-156) */
  OKVIS_BA_ARR_PAIR_W = 7,      /* [n_pair][18] W_(block,lm) = sum J_block^T J_l, row-major 6x3          */
  OKVIS_BA_ARR_REDUCED_S = 8,   /* [D][D]       damped reduced-camera matrix actually factorised        */
  OKVIS_BA_ARR_REDUCED_RHS = 9, /* [D]          reduced right-hand side (= -gradient after Schur)        */
  OKVIS_BA_ARR_STEP = 10,       /* [D]          last reduced step delta                                  */
  OKVIS_BA_ARR_LM_QUALITY = 11, /* [n_lm]       sqrt(lambda_min)/sqrt(lambda_max) (Estimator.cpp:880-896)*/
  OKVIS_BA_ARR_GRADIENT = 12,   /* [D]          un-reduced gradient of the pose/speed-bias part          */
  OKVIS_BA_ARR_IMU_RESIDUAL = 13,/* [n_imu][15] weighted IMU residual at the accepted state             */
  OKVIS_BA_ARR_HPP = 14,        /* [D][D]       un-reduced, un-damped pose/speed-bias Hessian block U (oracle only) */
  OKVIS_BA_ARR_DAMPING = 15,    /* [D]          clamp(diag U) used for the LM damping of the last solve   */
  OKVIS_BA_ARR_IMU_SB_REF = 16  /* [n_imu][9]   reference speed/bias of each ImuError's preintegration cache */
};

/* ---- lifecycle ----------------------------------------------------------------------------------- */
int okvis_ba_abi_version(void);
void okvis_ba_get_limits(okvis_ba_limits* out);
void okvis_ba_default_options(okvis_ba_options* out);
const char* okvis_ba_error_string(int status);

/* replaces `new okvis::ceres::Map` + ::ceres::Problem construction (Map.cpp:54-62).  device = HIP ordinal.
 * Fails with OKVIS_BA_ERR_NO_DEVICE when no GPU is visible: there is no CPU path. */
int okvis_ba_create(okvis_ba_solver** out, int device);
int okvis_ba_destroy(okvis_ba_solver* s);

/* replaces the sequence of Map::addParameterBlock / addResidualBlock / setParameterBlockConstant calls
 * (Map.cpp:292-434,568-576) that Estimator::addStates/addLandmark/addObservation issue: uploads the whole
 * structure + values of n independent windows and builds the device index arrays. */
int okvis_ba_upload(okvis_ba_solver* s, int n_windows, const okvis_ba_window* windows);
/* host-only structure check: runs exactly the index building of okvis_ba_upload (validation, reduced
 * ordering, (landmark,block) pairs, linearise groups, Schur chunks) without touching a device and without
 * any numeric work.  stats[8] = {D, Dp, n_pair, n_group, n_chunk, n_task, gpart doubles, arena bytes}.
 * opt may be NULL (defaults). */
int okvis_ba_check_window(const okvis_ba_window* w, const okvis_ba_options* opt, int64_t* stats);
/* The index lists that build would hand to the device, for a window that is one of n_windows sharing it (the batch size sets
 * the landmarks per linearise group and per Schur chunk): host only, diagnostics and tests.  `which` names a list
 * (OKVIS_BA_LIST_*); its entries are written to out as int32 (16-bit lists widened, records as consecutive ints: a group is
 * 16 ints {lm, obs, pair, task, pair-list, task-list begin/end, first piece, pieces before waves 1..3}, a task 6 {type,
 * offset a, offset b, list begin, list end, output}, a chunk 2 {group begin, end}).  *n = number of ints of the list;
 * nothing is written when capacity < *n (returns OKVIS_BA_ERR_ARG then, with *n set: ask with capacity 0 first). */
#define OKVIS_BA_LIST_GROUPS 0
#define OKVIS_BA_LIST_LM_OBS_BEGIN 1
#define OKVIS_BA_LIST_LM_PAIR_BEGIN 2
#define OKVIS_BA_LIST_PAIR_LM 3
#define OKVIS_BA_LIST_PAIR_BLOCK 4      /* piece path only */
#define OKVIS_BA_LIST_PAIR_OFF 5
#define OKVIS_BA_LIST_PAIR_ROLE 6
#define OKVIS_BA_LIST_LM_PIECE_BEGIN 7  /* piece path only */
#define OKVIS_BA_LIST_PAIR_PIECE 8      /* piece path only */
#define OKVIS_BA_LIST_PAIR_LIST_BEGIN 9
#define OKVIS_BA_LIST_PAIR_LIST 10      /* staged path only */
#define OKVIS_BA_LIST_TASKS 11
#define OKVIS_BA_LIST_TASK_LIST 12
#define OKVIS_BA_LIST_CHUNKS 13
#define OKVIS_BA_LIST_CHUNK_DIAG_BEGIN 14
#define OKVIS_BA_LIST_CHUNK_DIAG_OUT 15
#define OKVIS_BA_LIST_CHUNK_DESC 16
#define OKVIS_BA_LIST_PIECE_PATH 17     /* one int: 1 = the lists are those of the piece path (ba_linearize2.hpp) */
#define OKVIS_BA_LIST_LDL_COMP 18       /* one int: bit b = diagonal block b of the dense solver is eliminated with compensated products (it holds
                                           columns of a pose prior or of the marginalisation prior; 0 for windows above the LDS solver's size) */
#define OKVIS_BA_LIST_CHAIN 19          /* one int: > 0 = the window (as a batch of its own) is laid out for the chain solver (okvis_ba_tuning::
                                           solve_mode, ba_chain.hpp), the value is the number of speed/bias blocks; 0 = dense LDL^T.  The mask of
                                           OKVIS_BA_LIST_LDL_COMP then counts the 16-blocks of the POSE system                         */
int okvis_ba_check_window_lists(const okvis_ba_window* w, const okvis_ba_options* opt, int32_t n_windows, int32_t which,
                                int32_t* out, int64_t capacity, int64_t* n);
/* ---- incremental structure updates ---------------------------------------------------------------------
 * The reference edits its problem in O(1) per block: Map::addParameterBlock / addResidualBlock / removeResidualBlock /
 * removeParameterBlock (Map.cpp:292-565), driven by Estimator::addStates / addLandmark / addObservation / removeObservation /
 * applyMarginalizationStrategy (Estimator.cpp:110-365, 368-413, 434-773).  An okvis_ba_patch is one batch of such edits to a
 * window, applied in this order:
 *   1. removals, by index into the window as it stands (every list strictly ascending).  Removing a parameter block removes the
 *      terms attached to it, like Map::removeParameterBlock (Map.cpp:352-379): observations of a removed landmark / pose /
 *      extrinsics block, IMU terms and pose / speed-bias / relative-pose priors on a removed block.  The dense marginalisation
 *      prior cannot lose one of its blocks: replace it in the same patch (OKVIS_BA_PATCH_MARG_PRIOR), else OKVIS_BA_ERR_ARG;
 *   2. what is left is renumbered by stable compaction (relative order kept);
 *   3. appended blocks take the next indices (landmarks: or the places add_lm_before names); appended terms and replaced
 *      prior families are given in the NEW numbering;
 *      observations stay sorted by (lm, pose, cam): appended ones are merged in, behind equal keys;
 *   4. sparse value updates (new numbering), e.g. re-triangulated landmarks (Estimator::setLandmark).
 * A patch is checked completely before anything changes: on an error the window is untouched. */
#define OKVIS_BA_PATCH_POSE_PRIORS 1 /* bits of okvis_ba_patch::replace: the family is replaced wholesale by the patch's arrays */
#define OKVIS_BA_PATCH_SB_PRIORS 2
#define OKVIS_BA_PATCH_RELPOSE 4
#define OKVIS_BA_PATCH_MARG_PRIOR 8
typedef struct okvis_ba_patch {
  int32_t n_remove_obs;  const int32_t* remove_obs;
  int32_t n_remove_lm;   const int32_t* remove_lm;
  int32_t n_remove_pose; const int32_t* remove_pose;
  int32_t n_remove_sb;   const int32_t* remove_sb;
  int32_t n_remove_imu;  const int32_t* remove_imu;
  int32_t n_add_pose; const double* add_pose; /* [n_add_pose][7] */ const uint8_t* add_pose_fixed;
  int32_t n_add_sb;   const double* add_sb;   /* [n_add_sb][9]   */ const uint8_t* add_sb_fixed;
  int32_t n_add_lm;   const double* add_lm;   /* [n_add_lm][4]   */
  int32_t n_add_obs;  /* arrays as okvis_ba_window::obs_*, any order */
  const int32_t* add_obs_lm; const int32_t* add_obs_pose; const int32_t* add_obs_ext; const int32_t* add_obs_cam;
  const double* add_obs_uv; const double* add_obs_sqrtw;
  int32_t n_add_imu;  /* arrays as okvis_ba_window::imu_*; add_imu_s_begin indexes the patch's own sample arrays */
  const int32_t* add_imu_pose0; const int32_t* add_imu_sb0; const int32_t* add_imu_pose1; const int32_t* add_imu_sb1;
  const int64_t* add_imu_t0; const int64_t* add_imu_t1; const int32_t* add_imu_s_begin; const int32_t* add_imu_s_count;
  int32_t n_add_imu_samples;
  const int64_t* add_imu_s_t; const double* add_imu_s_gyr; const double* add_imu_s_acc;
  int32_t replace;    /* OKVIS_BA_PATCH_* bits; the arrays below are read only for the families named here */
  int32_t n_pprior;  const int32_t* pprior_pose; const double* pprior_meas; const double* pprior_sqrtinfo;
  int32_t n_sbprior; const int32_t* sbprior_sb;  const double* sbprior_meas; const double* sbprior_sqrtinfo;
  int32_t n_relpose; const int32_t* rel_pose0; const int32_t* rel_pose1; const double* rel_sqrtinfo;
  int32_t marg_dim; int32_t marg_nblocks; const int32_t* marg_block_type; const int32_t* marg_block_idx; const int32_t* marg_block_off;
  const double* marg_J; const double* marg_e0; const double* marg_lin;
  int32_t n_set_pose; const int32_t* set_pose_idx; const double* set_pose; /* [n_set_pose][7] */
  int32_t n_set_sb;   const int32_t* set_sb_idx;   const double* set_sb;   /* [n_set_sb][9]   */
  int32_t n_set_lm;   const int32_t* set_lm_idx;   const double* set_lm;   /* [n_set_lm][4]   */
  /* optional (NULL = every appended landmark takes the next index): add_lm_before[k] = how many of the landmarks that STAY come
   * in front of appended landmark k, ascending (0 .. number that stay).  Landmark k is then inserted there instead of at the
   * end, e.g. to keep the landmarks in the order of their ids like a flattened okvis::PointMap; landmark indices of the
   * appended observations and of set_lm_idx are the resulting ones. */
  const int32_t* add_lm_before;
} okvis_ba_patch;

/* Host-side window container with these edits (no device needed): create = deep copy of a window, patch = the edit above,
 * view = the window as okvis_ba_upload takes it (pointers valid until the next patch / destroy of this store). */
typedef struct okvis_ba_window_store okvis_ba_window_store;
int okvis_ba_store_create(const okvis_ba_window* w, okvis_ba_window_store** out);
int okvis_ba_store_patch(okvis_ba_window_store* st, const okvis_ba_patch* p);
int okvis_ba_store_view(const okvis_ba_window_store* st, okvis_ba_window* out);
void okvis_ba_store_destroy(okvis_ba_window_store* st);

/* A patchable solver keeps such a container of every window it uploads (set before okvis_ba_upload; default off: no copy).
 * okvis_ba_patch_window edits window w in place of a re-flatten + re-upload by the caller: the blocks that stay keep the values the
 * DEVICE holds (the accepted state of the last optimisation, the bias every IMU term's preintegration was last built at), the
 * index is rebuilt and the arena re-filled from the container.  Results are bit-identical to uploading okvis_ba_store_view of
 * the same edits with those values.  All or nothing: a patch that is rejected (OKVIS_BA_ERR_ARG) or whose result exceeds a structure
 * limit of okvis_ba_upload (that status) leaves the solver with the window it had, values included.
 * OKVIS_BA_ERR_STATE: solver not patchable or nothing uploaded. */
int okvis_ba_set_patchable(okvis_ba_solver* s, int on);
int okvis_ba_patch_window(okvis_ba_solver* s, int w, const okvis_ba_patch* p);
/* the container of window w of a patchable solver as it stands (pointers valid until the next upload / patch) */
int okvis_ba_patched_view(okvis_ba_solver* s, int w, okvis_ba_window* out);
/* The NUMBERS of window w's marginalisation prior once more — J [marg_dim][marg_dim] row-major, e0 [marg_dim] — for the blocks,
 * offsets and linearisation points the window was uploaded or patched with (they stay as they are).  For a caller whose prior
 * is still being computed when the window is handed over (okvis_ba_marginalize_begin on another solver): it uploads / patches with
 * the block structure and stand-in numbers, waits for the numbers meanwhile, and sets them here — before the next
 * okvis_ba_begin / optimize.  One copy; a patchable solver's container is updated too.  OKVIS_BA_ERR_ARG if the window has no
 * prior. */
int okvis_ba_set_marg_prior_values(okvis_ba_solver* s, int w, const double* J, const double* e0);

/* overwrite only block VALUES of window w (Estimator::set_T_WS/setSpeedAndBias/setLandmark,
 * Estimator.cpp:1205-1302); any pointer may be NULL to keep the device copy. */
int okvis_ba_set_state(okvis_ba_solver* s, int w, const double* pose, const double* sb, const double* lm);
int okvis_ba_set_options(okvis_ba_solver* s, const okvis_ba_options* opt);

/* ---- the hot path ---------------------------------------------------------------------------------- */
/* replaces Estimator::optimize(numIter, numThreads, verbose) -> Map::solve() -> ::ceres::Solve
 * (Estimator.cpp:843-877, Map.hpp:371-373) for every uploaded window at once.  summaries may be NULL or
 * [n_windows].  Also refreshes landmark quality (Estimator.cpp:880-900). */
int okvis_ba_optimize(okvis_ba_solver* s, int num_iter, okvis_ba_summary* summaries);

/* time-limited variant (CeresIterationCallback.hpp:77-86 + Estimator::setOptimizationTimeLimit,
 * Estimator.cpp:909-929): always runs min_iter iterations, then stops as soon as elapsed wall time
 * exceeds time_limit_s (time_limit_s < 0: no limit). */
int okvis_ba_optimize_timed(okvis_ba_solver* s, int max_iter, int min_iter, double time_limit_s,
                            okvis_ba_summary* summaries);

/* step-wise entry points (one kernel sequence each; what one trust-region iteration is made of).
 *   begin:      evaluate + linearise all factors at the uploaded state, accept it (iteration 0)
 *   iterate(n): n x [Schur reduce -> reduced solve -> back-substitute + (+)update + re-linearise]
 *   finish:     download nothing; finalise summaries + landmark quality
 * okvis_ba_optimize == begin + iterate(num_iter) + finish. */
int okvis_ba_begin(okvis_ba_solver* s);
int okvis_ba_iterate(okvis_ba_solver* s, int n);
int okvis_ba_finish(okvis_ba_solver* s, okvis_ba_summary* summaries);

/* cost-only evaluation at the accepted state: 0.5*sum rho(|r|^2) (Ceres Evaluate(params,r,NULL) path,
 * implementation/ReprojectionError.hpp:127-132, ImuError.cpp:608). costs = [n_windows]. */
int okvis_ba_evaluate_cost(okvis_ba_solver* s, double* costs);

/* ---- results -------------------------------------------------------------------------------------- */
/* Estimator::get_T_WS / getSpeedAndBias / getLandmark (Estimator.cpp:933-1200): copy the accepted state
 * of window w back; any pointer may be NULL. */
int okvis_ba_get_state(okvis_ba_solver* s, int w, double* pose, double* sb, double* lm);
/* Everything Estimator::optimize copies back after the solve (Estimator.cpp:880-906: landmark quality + estimates;
 * plus the bias each ImuError cache was (re)built at, which the reference keeps inside the ImuError object) with ONE
 * stream synchronisation: pose [n_pose][7], sb [n_sb][9], lm [n_lm][4], lm_quality [n_lm] (OKVIS_BA_ARR_LM_QUALITY),
 * imu_sb_ref [n_imu][9] (OKVIS_BA_ARR_IMU_SB_REF); any pointer may be NULL. */
int okvis_ba_fetch_results(okvis_ba_solver* s, int w, double* pose, double* sb, double* lm, double* lm_quality,
                           double* imu_sb_ref);
/* The preintegration records of window w's IMU terms, caches [n_imu][OKVIS_BA_IMU_CACHE_DOUBLES], from the same packed record
 * as okvis_ba_fetch_results (no further synchronisation after it): what okvis_ba_window::imu_cache takes back with flag 2.  A term
 * that has not been evaluated since the upload has nothing to hand out: its record's flag word is 0 and it must travel with
 * flag 0 or 1. */
int okvis_ba_fetch_imu_caches(okvis_ba_solver* s, int w, double* caches);
/* size (in doubles) and contents of an intermediate array of window w (parity tests) */
int okvis_ba_array_size(okvis_ba_solver* s, int w, int which, int64_t* n_doubles);
int okvis_ba_download(okvis_ba_solver* s, int w, int which, double* out, int64_t n_doubles);
/* structure queries: reduced dimension and (block,landmark) pair list as built by upload */
int okvis_ba_reduced_dim(okvis_ba_solver* s, int w, int32_t* dim);
int okvis_ba_pair_count(okvis_ba_solver* s, int w, int32_t* n_pair);
int okvis_ba_pairs(okvis_ba_solver* s, int w, int32_t* pair_lm, int32_t* pair_block);

/* ---- measurement hooks (bench.py) ----------------------------------------------------------------- */
/* HIP-event timing of the NEXT okvis_ba_iterate call on the solver's own stream: after the call,
 * total_ms = elapsed between events bracketing the n iterations. */
int okvis_ba_last_iterate_ms(okvis_ba_solver* s, float* total_ms);
/* per-kernel HIP-event timing (eager launches, events around every kernel): fills ms[4] with the summed
 * time of the {Schur, solve, IMU/prior, linearise} kernels over n iterations. Slower than graph replay;
 * used only to attribute time for the roofline object. */
int okvis_ba_profile_iterations(okvis_ba_solver* s, int n, float* ms4);
/* the same pass with one record per iteration: ms[i][4] = durations of the {Schur, solve, -, linearise} launches of
 * iteration i (bench.py reports the median launch and the launches that carry an IMU re-preintegration separately) */
int okvis_ba_profile_launches(okvis_ba_solver* s, int n, float* ms /* [n][4] */);
/* algorithmic bytes one iteration moves for the uploaded batch (formula in DESIGN.md §4) */
int okvis_ba_algorithmic_bytes(okvis_ba_solver* s, int64_t* linearize_bytes, int64_t* schur_bytes,
                               int64_t* solve_bytes, int64_t* small_bytes);
int okvis_ba_synchronize(okvis_ba_solver* s);
/* Launches of few windows sum the Schur partials on helper workgroups next to the solving one.  A solving
 * workgroup whose helpers are late (not co-scheduled) sums the partials itself, with the same result, and counts it:
 * count = such time-outs over all windows since the upload.  0 in a healthy run; tests and bench.py report it. */
int okvis_ba_helper_timeouts(okvis_ba_solver* s, int64_t* count);
/* Which launches the uploaded batch takes under the solver's current options (read-only, nothing is launched): the parity tests of
 * the headline configuration assert the route they mean to test (tests/test_gpu_batch64.py).  route[OKVIS_BA_ROUTE_COUNT]. */
#define OKVIS_BA_ROUTE_WINDOWS 0
#define OKVIS_BA_ROUTE_FUSED 1                  /* 1 = linearise + landmark reduction in one launch, no Schur launch                */
#define OKVIS_BA_ROUTE_DECISION_FREE_SCHUR 2    /* 1 = the Schur launch reduces the trial buffer, the solve kernel decides          */
#define OKVIS_BA_ROUTE_PIECE_PATH 3             /* 1 = linearize2_kernel                                                            */
#define OKVIS_BA_ROUTE_SPLIT_SMALL 4            /* 1 = IMU / prior factors in small_kernel, a launch of their own                   */
#define OKVIS_BA_ROUTE_SUB_BATCHES 5            /* streams the batch is spread over                                                 */
#define OKVIS_BA_ROUTE_SUB_BATCH_MAX_WINDOWS 6
#define OKVIS_BA_ROUTE_SCHUR_KERNEL 7           /* 0 none, 1 schur_kernel, 2 schur_mfma_kernel<3>, 3 schur_mfma_kernel<9>           */
#define OKVIS_BA_ROUTE_SOLVE_DBUF 8             /* 1 = solve_kernel<false, true> (one set of Schur partials per linearisation buffer) */
#define OKVIS_BA_ROUTE_SOLVE_TILED 9            /* 1 = at least one window above the LDS solver's size (chol_tiles_window_kernel)    */
#define OKVIS_BA_ROUTE_SOLVE_HELPERS 10         /* helper workgroups per solving workgroup (0 above 8 windows per launch)           */
#define OKVIS_BA_ROUTE_GRAPH 11
#define OKVIS_BA_ROUTE_MAX_CHUNKS 12            /* Schur chunks of the window that has most                                         */
#define OKVIS_BA_ROUTE_SLOTS 13                 /* launch slots since okvis_ba_begin                                                */
#define OKVIS_BA_ROUTE_SOLVE_MODE 14            /* OKVIS_BA_SOLVE_* the LDS-resident windows are solved with                         */
#define OKVIS_BA_ROUTE_SMALL_RIDES 15           /* 1 = the IMU / prior factors are evaluated inside the Schur launch (schur_ride_kernel),
                                                   small_prepare_kernel behind the solve launch only keeps the preintegrations up to date */
#define OKVIS_BA_ROUTE_COUNT 16
int okvis_ba_launch_route(okvis_ba_solver* s, int32_t* route);

/* ---- marginalisation -------------------------------------------------------
 * Numeric core of okvis::Estimator::applyMarginalizationStrategy (Estimator.cpp:434-773), i.e. what the
 * reference's MarginalizationError does between the first addResidualBlock and updateErrorComputation
 * (MarginalizationError.cpp:127-435 linearise-and-accumulate, :507-802 marginalizeOut, :806-846
 * updateErrorComputation).  The caller uploads, as an ordinary window, exactly the residuals that are to be
 * linearised into the prior, with every block's VALUE set to its linearisation point (first-estimate
 * Jacobians, :292-310): all landmarks of that window are eliminated (landmark path, :617-684, per-block
 * preconditioned pseudo-inverse), then the flagged pose-type / speed-bias blocks (dense path, :686-739), and
 * the remaining system is turned into the error-term form J, e0 (:806-846).  The previous prior enters as
 * (H, b0) over blocks of this window (the reference keeps H_ and b0_ inside the object).  The window itself
 * must not carry a marg_* prior.  Reduced dimension of the window <= OKVIS_BA_MARG_MAX_WINDOW_DIM (= okvis_ba_limits::
 * max_reduced_dim) and previous prior <= okvis_ba_limits::max_marg_dim rows; sub-windows of up to 174 reduced dimensions with
 * a prior of up to 192 rows take the path whose matrices stay in LDS, larger ones the same arithmetic in an HBM workspace. */
#define OKVIS_BA_MARG_MAX_WINDOW_DIM 900
typedef struct okvis_ba_marg_spec {
  const uint8_t* pose_marg;          /* [n_pose] 1 = eliminate this (free) pose-type block */
  const uint8_t* sb_marg;            /* [n_sb]   1 = eliminate this (free) speed/bias block */
  int32_t prior_dim, prior_nblocks;  /* previous prior (0 = none) */
  const int32_t* prior_block_type;   /* OKVIS_BA_BLOCK_POSE / OKVIS_BA_BLOCK_SPEEDBIAS */
  const int32_t* prior_block_idx;    /* block index in this window */
  const int32_t* prior_block_off;    /* first row/column of the block inside prior_H */
  const double* prior_H;             /* [prior_dim][prior_dim] row-major (H_) */
  const double* prior_b0;            /* [prior_dim] (b0_) */
} okvis_ba_marg_spec;

typedef struct okvis_ba_marg_result {
  int32_t capacity_dim, capacity_blocks;  /* in: sizes of the caller-allocated arrays below */
  int32_t dim, nblocks, rank;             /* out: size of the new prior, its blocks, numeric rank of H */
  int32_t* block_type;                    /* [capacity_blocks] */
  int32_t* block_idx;                     /* [capacity_blocks] index in the uploaded window */
  int32_t* block_off;                     /* [capacity_blocks] */
  double* H;                              /* [dim][dim] row-major, packed with leading dimension dim */
  double* b0;                             /* [dim] */
  double* J;                              /* [dim][dim]: J^T J = H up to the dropped eigenvalues; defined up to an
                                             orthogonal transformation of its rows (eigen form, or a (pivoted)
                                             Cholesky factor with zero rows below the rank) */
  double* e0;                             /* [dim] */
  int32_t sweeps[2];                      /* out (diagnostic): Jacobi sweeps of the two decompositions (0 = a Cholesky
                                             path decided the rank) */
} okvis_ba_marg_result;

int okvis_ba_marginalize(okvis_ba_solver* s, int w, const okvis_ba_marg_spec* spec, okvis_ba_marg_result* result);
/* The same call in two halves, for a caller that has work of its own to do while the device computes (the reference's
 * applyMarginalizationStrategy goes on to delete what was marginalised, Estimator.cpp:694-745, and the frontend adds the next frame
 * before the prior is read again).  _begin checks the arguments, copies what it needs of `spec`, enqueues everything and returns
 * with the blocks the new prior connects (result->dim, nblocks, block_type / block_idx / block_off: known without the numbers);
 * _end waits and fills H, b0, J, e0, rank, sweeps of a result with the same capacities (it may be the same struct).  Between
 * the two the solver takes no edits and hands out no results: okvis_ba_upload, _patch_window, _set_state, _get_state,
 * _set_marg_prior_values, _begin, _iterate, _finish (and with them _optimize / _optimize_timed / _evaluate_cost), _fetch_results,
 * _fetch_imu_caches, _download and another _marginalize / _marginalize_begin all return OKVIS_BA_ERR_STATE; only the queries that
 * touch neither the device nor the window (_reduced_dim, _pair_count, _get_limits, ...) and _synchronize are served.  Sizes beyond
 * the LDS route are waited for in _begin (their fall-back needs the call's arguments).  _end looks at the result structure before
 * it ends the call: with too little room (OKVIS_BA_ERR_ARG) nothing is lost and _end can be called again with more.  A numeric
 * failure is reported by _end (OKVIS_BA_ERR_NUMERIC). */
int okvis_ba_marginalize_begin(okvis_ba_solver* s, int w, const okvis_ba_marg_spec* spec, okvis_ba_marg_result* result);
int okvis_ba_marginalize_end(okvis_ba_solver* s, okvis_ba_marg_result* result);

/* ---- multi-GPU driver --------------------------------------------------------------
 * Windows are independent units: window i of a job runs on rank i mod world, one process per GPU, no data-path
 * collective.  The only exchange is ONE all-gather of these fixed-size records after all windows finished (RCCL over
 * xGMI through torch.distributed in bench.py / okvis_amd/dist.py; the launcher owns the communicator, this library the
 * sharding rule, the run and the records). */
typedef struct okvis_ba_window_record {
  uint32_t window_id;   /* index of the window in the job */
  uint32_t iterations;  /* trust-region iterations performed */
  double final_cost;
  double seconds;       /* wall time of the rank's batch that contained this window */
} okvis_ba_window_record;
/* ids_out[*n_out] = the windows of rank `rank` (capacity: (n_total + world - 1) / world entries) */
int okvis_ba_shard(int32_t n_total, int32_t rank, int32_t world, int32_t* ids_out, int32_t* n_out);
/* shard -> upload the rank's windows on `device` -> optimize(num_iter) -> one record per local window.
 * all_windows: the n_total windows of the job (only the rank's share is touched).  records_out capacity as above. */
int okvis_ba_batch_run(int device, int32_t rank, int32_t world, int32_t n_total, const okvis_ba_window* all_windows,
                       const okvis_ba_options* opt, int num_iter, okvis_ba_window_record* records_out, int32_t* n_out);
/* The all-gather of the records WITHOUT a Python launcher in the data path: RCCL loaded at run time (dlopen librccl.so;
 * OKVIS_BA_RCCL_LIB names another file), one communicator per call.  Every rank passes n_per_rank records (ranks with
 * fewer windows pad with window_id = 0xffffffff); all [world][n_per_rank].  The ncclUniqueId goes from rank 0 to the others
 * through `id_file` (a path every rank sees; written with an atomic rename, polled for up to timeout_s; not needed for
 * world = 1).  The path belongs to ONE gather: rank 0 removes a left-over file before it publishes and removes its own as soon as
 * the communicator stands (or it fails); two jobs must not share a path, and ranks > 0 must not be started against the file of an
 * earlier job that rank 0 has not yet replaced.  OKVIS_BA_ERR_UNSUPPORTED: no RCCL library found (checked on every rank before an
 * id changes hands); OKVIS_BA_ERR_STATE: id file timeout or an RCCL error. */
int okvis_ba_gather_records(int32_t rank, int32_t world, int device, const char* id_file, double timeout_s,
                            const okvis_ba_window_record* mine, int32_t n_per_rank, okvis_ba_window_record* all);
/* okvis_ba_batch_run + okvis_ba_gather_records: every rank returns the records of ALL n_total windows in window order */
int okvis_ba_batch_run_gathered(int device, int32_t rank, int32_t world, int32_t n_total, const okvis_ba_window* all_windows,
                                const okvis_ba_options* opt, int num_iter, const char* id_file, double timeout_s,
                                okvis_ba_window_record* all_records);

/* ---- diagnostics ------------------------------------------------------------------------------------
 * The dense solver behind windows whose reduced dimension exceeds the single-workgroup LDS path
 * (OKVIS_BA_MARG_MAX_WINDOW_DIM): tiled multi-workgroup Cholesky with fp64 MFMA tile updates
 * (okvis_amd/csrc/ba_chol_tiles.hpp), exposed stand-alone for tests and profiling.  Solves S x = rhs for a
 * symmetric positive definite S [n][n] (row-major, full storage).  *info = 0 ok, 1 = not positive definite /
 * dependency timeout. */
int okvis_ba_dense_solve(int device, int32_t n, const double* S, const double* rhs, double* x, int32_t* info);
/* One reduced camera system through the LDS-resident solvers of the solve kernel, on its own (unit tests and A/B timing of the
 * two solvers; D <= the LDS solver's limit, (D - Dp) a multiple of 9).  S: D x D symmetric, row-major; Dp: rows of the pose part
 * (pose-type blocks first, then 9-row speed/bias blocks).  mode: OKVIS_BA_SOLVE_DENSE = blocked LDL^T of the whole system,
 * OKVIS_BA_SOLVE_CHAIN = speed/bias blocks eliminated along the chain first — S must then have the chain structure (a speed/bias
 * block couples only to its two neighbours and to poses).  comp_mask: diagonal 16-blocks eliminated with compensated products (of
 * the whole system / of the pose system).  ticks: device clock ticks of one solve (mean over `repeats`); info: 1 = a pivot was not
 * positive.  lds_dump (NULL or lds_capacity doubles, at most 20 000 are needed): the solver's LDS image after the solve. */
int okvis_ba_reduced_solve(int device, int32_t D, int32_t Dp, int32_t mode, uint32_t comp_mask, const double* S, const double* rhs,
                           double* x, int64_t* ticks, int32_t* info, int32_t repeats, double* lds_dump, int64_t lds_capacity);

#ifdef __cplusplus
}
#endif
#endif /* OKVIS_AMD_BA_H_ */
