/* fgo.h — C-ABI of the MI355X-native batch factor-graph optimiser (libfgo.so).
 *
 * This is the drop-in boundary for the optimiser back-end that rising-turtle/graph_slam delegates
 * to g2o / GTSAM.  Every entry point names the reference call site it replaces
 * (paths relative to the reference tree).  Plain C: opaque context, pointers and sizes, no C++ or
 * torch types.  Every function returns 0 on success or a negative FGO_E* code and never throws;
 * fgo_last_error() gives the message.  One context per caller thread (the reference's wrappers
 * are single-threaded too: g2o/g2o_graph.cpp, gtsam/gtsam_graph.cpp).  Arrays passed in are
 * copied; the context owns all device (HBM) memory.  All arithmetic is IEEE f64.
 *
 * Conventions: pose = t[3] + unit quaternion q[4] in (x,y,z,w) order (Eigen coeff order);
 * X = (R(q), t) maps local -> world.  Information matrices are passed as the 21 upper-triangular
 * entries, row-major (O00 O01 .. O05 O11 ..), the VRO record order (gtsam/gtsam_graph.cpp:1574-1590).
 */
#ifndef FGO_H
#define FGO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libfgo.so is built with -fvisibility=hidden: the declarations below are the whole export list. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define FGO_OK 0
#define FGO_EINVAL (-1)   /* bad argument / unknown id                                   */
#define FGO_ENODEV (-2)   /* no HIP device, or the HIP runtime reported an error          */
#define FGO_ENOMEM (-3)
#define FGO_ESTATE (-4)   /* nothing to optimise (g2o's optimize() == -1)                */
#define FGO_ENUM (-5)     /* numerical failure: factorisation not positive definite       */

/* tangent / information ordering of an SE3 edge */
#define FGO_TANGENT_G2O 0    /* [t; q]   g2o EdgeSE3        (g2o/g2o_graph.cpp:125-132)      */
#define FGO_TANGENT_GTSAM 1  /* [w; v]   gtsam BetweenFactor (gtsam/gtsam_graph.cpp:689-692) */

typedef struct fgo_ctx fgo_ctx;

typedef struct {
  int device;          /* HIP device ordinal (default 0)                                   */
  int verbose;         /* 0 silent (reference: setVerbose(false), g2o_graph.cpp:70)        */
  int ordering;        /* 0 = nested dissection + local minimum degree (default)           */
  int nd_leaf;         /* nested-dissection leaf size in poses (0 = default 64)            */
  int order_candidates;/* 0 / 1: one ordering (default).  2 .. 4: that many orderings (balance weight / leaf size variants) are built
                          and the one with the lowest predicted sweep time (levels x 76 us + block updates x 0.051 ns) is kept:
                          +2 .. 4 % iterations/s on 100k-pose graphs for 2 .. 4 x the structure phase -- for long runs      */
  int reserved[11];
} fgo_config;

/* Result of one fgo_optimize() call == one g2o SparseOptimizer::optimize(n) call. */
typedef struct {
  int iterations;      /* LM iterations performed (what optimize() returns)                */
  int trials;          /* linear solves, accepted + rejected                               */
  int terminated;      /* algorithm returned 'Terminate' (10 failed trials / rho == 0)     */
  int structure_rebuilt;
  double chi2_initial, chi2_final, lambda_final;
  /* wall-clock (host) seconds */
  double t_symbolic, t_upload, t_total;
  /* device milliseconds from hipEvents on the context's stream */
  double ms_linearize, ms_factor, ms_solve, ms_update;
  /* structure */
  int64_t n_free, n_edges, nnz_H_blocks, nnz_L_blocks, n_update_ops;
  int n_levels, n_tasks;
  /* algorithmic HBM bytes of ONE pass of each phase: factor = H read + L written + L re-read by the
     forward solve fused into the factor sweep; solve = the backward sweep (L read once) */
  double bytes_factor, bytes_linearize, bytes_solve;
  double reserved[8];
} fgo_stats;

/* ---- lifecycle: replaces CGraphG2O::createOptimizer (g2o/g2o_graph.cpp:65-77) and the
 *      NonlinearFactorGraph/Values allocation in CGraphGT::CGraphGT (gtsam/gtsam_graph.cpp:75-91) */
fgo_ctx *fgo_create(const fgo_config *cfg);          /* NULL cfg = defaults; NULL on failure */
void fgo_destroy(fgo_ctx *ctx);                      /* ~CGraphG2O: g2o_graph.cpp:50-56       */
const char *fgo_last_error(const fgo_ctx *ctx);      /* ctx may be NULL (creation errors)     */
const char *fgo_version(void);
int fgo_device_count(void);                          /* HIP devices visible, <0 on error      */

/* ---- variables: VertexSE3 creation, g2o/g2o_graph.cpp:88-91 (fixed first vertex), :115-119;
 *      Values::insert / update of Pose3, gtsam/gtsam_graph.cpp:331,655-669 */
int fgo_add_pose(fgo_ctx *ctx, int64_t id, const double t[3], const double q_xyzw[4], int fixed);
int fgo_set_pose(fgo_ctx *ctx, int64_t id, const double t[3], const double q_xyzw[4]);
int fgo_set_fixed(fgo_ctx *ctx, int64_t id, int fixed);       /* OptimizableGraph::Vertex::setFixed ("FIX id" of a .g2o file) */
int fgo_get_pose(fgo_ctx *ctx, int64_t id, double out7[7]);   /* VertexSE3::estimate(), :297,327 */
int fgo_has_pose(const fgo_ctx *ctx, int64_t id);             /* mp_optimizer->vertex(id) != 0, :98-99 */
int64_t fgo_num_poses(const fgo_ctx *ctx);
int64_t fgo_num_edges(const fgo_ctx *ctx);
/* bulk forms (poses7 = n x 7, ids may be NULL for 0..n-1 continuing from the current count) */
int fgo_add_poses(fgo_ctx *ctx, int64_t n, const int64_t *ids, const double *poses7,
                  const unsigned char *fixed);
int fgo_get_poses(fgo_ctx *ctx, int64_t n, const int64_t *ids, double *poses7);

/* ---- factors: EdgeSE3 + setMeasurement + setInformation, g2o/g2o_graph.cpp:125-132;
 *      BetweenFactor<Pose3> + Gaussian::Information, gtsam/gtsam_graph.cpp:689-692 */
int fgo_add_edge_se3(fgo_ctx *ctx, int64_t id_i, int64_t id_j, const double t[3],
                     const double q_xyzw[4], const double info_ut21[21], int tangent_order);
int fgo_add_edges_se3(fgo_ctx *ctx, int64_t n, const int64_t *id_i, const int64_t *id_j,
                      const double *meas7, const double *info_ut21, int tangent_order);

/* ---- GTSAM-semantics factors.  A context is either a g2o-semantics graph (FGO_TANGENT_G2O edges, solved by
 *      fgo_optimize) or a GTSAM-semantics graph (FGO_TANGENT_GTSAM edges + priors, solved by fgo_optimize_gtsam);
 *      mixing the two in one context is an error (their Jacobians refer to different retractions).
 *      PriorFactor<Pose3>(X(id), mean, noise): gtsam/gtsam_graph.cpp:338-341 (Diagonal::Sigmas(1e-7 x 6) ->
 *      info = diag(1/sigma^2)); information passed as 21 upper-triangular entries in [omega; v] order. */
int fgo_add_prior_pose(fgo_ctx *ctx, int64_t id, const double t[3], const double q_xyzw[4], const double info_ut21[21]);
/* Plane landmarks: Values::insert(L(id), OrientedPlane3(a,b,c,d)) — gtsam/gtsam_graph.cpp:1198-1202 — and
 *      OrientedPlane3Factor(z, noiseModel::Gaussian::Covariance(S), X(pose), L(plane)) — gtsam/gtsam_graph.cpp:1265.
 *      z = measured plane (a,b,c,d) in the pose frame; cov_ut6 = upper triangle of the 3x3 covariance S, row-major.
 *      Any variable's current value is read back with fgo_get_pose (7 slots: plane = nx ny nz d, point = x y z). */
int fgo_add_plane(fgo_ctx *ctx, int64_t id, const double abcd[4]);
int fgo_add_plane_factor(fgo_ctx *ctx, int64_t pose_id, int64_t plane_id, const double z_abcd[4], const double cov_ut6[6]);
/* Bundle adjustment: Values::insert(Q(id), Point3) + PriorFactor<Point3>(Isotropic::Sigma(3, sigma)) —
 *      gtsam/gtsam_graph.cpp:379,387-394; Cal3DS2(fx,fy,s,u0,v0,k1,k2[,p1,p2]) — :373; GenericProjectionFactor<Pose3,
 *      Point3, Cal3DS2>(z, Isotropic::Sigma(2, sigma), X, Q, K, false, false, body_P_sensor) — :405-409.
 *      body_P_sensor7 = t(3) q_xyzw(4), NULL = identity. */
int fgo_add_point3(fgo_ctx *ctx, int64_t id, const double xyz[3]);
int fgo_add_prior_point3(fgo_ctx *ctx, int64_t id, const double xyz[3], double sigma);
int fgo_set_calib_ds2(fgo_ctx *ctx, double fx, double fy, double s, double u0, double v0, double k1, double k2, double p1,
                      double p2, const double body_P_sensor7[7]);
int fgo_add_reproj(fgo_ctx *ctx, int64_t pose_id, int64_t point_id, const double uv[2], double sigma);
/* bulk forms: n points (+ PriorFactor<Point3> when prior_sigma > 0) / n projection factors */
int fgo_add_points3(fgo_ctx *ctx, int64_t n, const int64_t *ids, const double *xyz, double prior_sigma);
int fgo_add_reprojs(fgo_ctx *ctx, int64_t n, const int64_t *pose_ids, const int64_t *point_ids, const double *uv, double sigma);
/* ---- IMU: velocity / bias variables, their priors, preintegration and the CombinedImuFactor.
 *      Values::insert(V(id), Vector3) / insert(B(id), imuBias::ConstantBias) + PriorFactor<Vector3>(Isotropic::Sigma(3,
 *      1e-3)) / PriorFactor<ConstantBias>(Isotropic::Sigma(6, 1e-3)) — gtsam/gtsam_graph.cpp:346-367.
 *      bias = [acc(3); gyro(3)].  fgo_preint mirrors PreintegratedCombinedMeasurements (on-manifold form; the
 *      reference's preintegration type is a GTSAM build flag, gtsam/imu_base.h:73 — see DESIGN.md):
 *      fgo_preint_reset + fgo_preint_integrate(acc, gyro, dt) replace resetIntegrationAndSetBias + the
 *      integrateMeasurement loop of CImuBase::predictNext (gtsam/imu_base.cpp:72-87; host-side, per factor);
 *      fgo_imu_params_vn100 = CImuVn100::getIMUParams + MakeSharedD(9.71) (gtsam/imu_vn100.cpp:24-67,
 *      gtsam/imu_base.cpp:258-263); fgo_add_imu_combined = CombinedImuFactor(X(i-1), V(i-1), X(i), V(i), B(i-1), B(i),
 *      preint) added to the graph (gtsam/test_ba_imu_graph.cpp:239-244). */
typedef struct {
  double dt;
  double dR[4];                /* preintegrated rotation, quaternion x y z w */
  double dp[3], dv[3];
  double J_R_bg[9], J_p_ba[9], J_p_bg[9], J_v_ba[9], J_v_bg[9];   /* row-major 3x3 bias Jacobians */
  double bhat[6];              /* bias the measurements were corrected with: acc(3), gyro(3) */
  double cov[225];             /* preintMeasCov, row-major 15x15, order theta p v bias_acc bias_gyro */
} fgo_preint;
typedef struct {
  double acc_cov, gyro_cov, integ_cov, bias_acc_cov, bias_gyro_cov, bias_acc_omega_int;   /* isotropic variances */
  double gravity[3];           /* n_gravity (navigation frame) */
} fgo_imu_params;
void fgo_imu_params_vn100(fgo_imu_params *p);
void fgo_preint_reset(fgo_preint *m, const double bias_hat6[6]);
void fgo_preint_integrate(fgo_preint *m, const fgo_imu_params *p, const double acc[3], const double gyro[3], double dt);
/* PreintegratedCombinedMeasurements::predict(state_i, bias_i): pose_j (7) and velocity_j (3) */
void fgo_preint_predict(const fgo_preint *m, const double gravity[3], const double pose_i7[7], const double vel_i[3],
                        const double bias_i6[6], double pose_j7[7], double vel_j[3]);
/* Batched preintegration on the GPU (the factors are independent; the reference runs the
 * integrateMeasurement loop of CImuBase::predictNext serially on the CPU, gtsam/imu_base.cpp:72-87).  Factor f integrates
 * the samples [sample_ptr[f], sample_ptr[f+1]) of acc / gyro (3 doubles per sample) with step dt, starting from
 * fgo_preint_reset(bias_hat6 + 6 f) (zero bias if NULL).  Same arithmetic as fgo_preint_integrate.  Host arrays in and
 * out; FGO_ENODEV without a HIP device (no CPU fallback -- use fgo_preint_integrate for that). */
int fgo_preint_batch(int device, int64_t n, const int64_t *sample_ptr, const double *acc, const double *gyro, double dt,
                     const double *bias_hat6, const fgo_imu_params *params, fgo_preint *out);
int fgo_add_vec3(fgo_ctx *ctx, int64_t id, const double xyz[3]);
int fgo_add_bias(fgo_ctx *ctx, int64_t id, const double bias6[6]);
int fgo_add_prior_vec3(fgo_ctx *ctx, int64_t id, const double xyz[3], double sigma);
int fgo_add_prior_bias(fgo_ctx *ctx, int64_t id, const double bias6[6], double sigma);
int fgo_set_gravity(fgo_ctx *ctx, const double n_gravity[3]);     /* default (0, 0, 9.71) */
int fgo_add_imu_combined(fgo_ctx *ctx, const int64_t ids6[6] /* Xi Vi Xj Vj Bi Bj */, const fgo_preint *preint);
/* the 15x15 information matrix (row-major, order theta p v ba bg) fgo_add_imu_combined gives the factor:
 * preintMeasCov^-1 by Cholesky, symmetrised -- noiseModel::Gaussian::Covariance(pim.preintMeasCov()) in GTSAM terms.
 * Host-only; FGO_ENUM if the covariance is not positive definite. */
int fgo_preint_information(const fgo_preint *preint, double info225[225]);

/* LevenbergMarquardtOptimizer(graph, values).optimize() with GTSAM 4.0's default parameters —
 *      CGraphGT::optimizeGraphBatch, gtsam/gtsam_graph.cpp:1784-1788.  max_iters <= 0 selects the default 100.
 *      Returns the number of iterations performed or a negative code. */
int fgo_optimize_gtsam(fgo_ctx *ctx, int max_iters, fgo_stats *stats /* may be NULL */);
/* ISAM2 semantics — CGraphGT::optimizeGraphIncremental, gtsam/gtsam_graph.cpp:1768-1776:
 *      isam2->update(new factors, new values);  values = isam2->calculateEstimate();
 *      with ISAM2Params{relinearizeThreshold (reference: 0.1), relinearizeSkip = 1} (:93-99) and the Gauss-Newton
 *      (undamped) step of ISAM2's default optimisation parameters.  "New" = everything added through fgo_add_* since the
 *      previous call.  The context keeps ISAM2's linearisation point theta and linear solution delta per variable; one
 *      call = { theta_v <- theta_v (+) delta_v, delta_v <- 0 for every variable with max|delta_v| >= threshold; linearise
 *      all factors at theta; solve H delta = b; values <- theta (+) delta }, i.e. what ISAM2's partial re-elimination
 *      computes with wildfireThreshold -> 0, evaluated as one full device sweep (the resident factorisation is rebuilt
 *      rather than edited; the structure phase reruns only when factors or variables were added).
 *      Returns 1, or a negative code (FGO_ENUM: system not positive definite; values, theta and delta are then as the
 *      relinearisation step left them).  stats->reserved[1] = number of variables relinearised; stats->reserved[3] = tasks of
 *      the elimination tree this update re-factored (-1: full sweep, -2: full sweep because most of the tree was affected);
 *      stats->reserved[4] = 1 if the back-substitution was cut by the wildfire threshold (fgo_isam2_set_wildfire), else 0.
 *      A structure built with landmarks eliminated (fgo_optimize_gtsam on a bundle-adjustment graph) cannot serve ISAM2:
 *      the first successful update switches the context to the generic form (one rebuild) until fgo_isam2_reset; a call
 *      that fails leaves that choice as it was. */
int fgo_isam2_update(fgo_ctx *ctx, double relinearize_threshold, fgo_stats *stats /* may be NULL */);
/* Growth reserve of the incremental mode.  A context that is driven through fgo_isam2_update builds its structure for
 * the graph PLUS `reserve_variables` phantom variables, each coupled to the `window` variables added before it: later
 * variables claim the phantom slots and later factors whose variable pairs lie inside that band (odometry, look-back,
 * IMU, plane and landmark factors of the newest key frames: gtsam/test_vro_imu_graph.cpp:159-350) are appended to the
 * device arrays in place -- no ordering, no symbolic factorisation, no re-upload; stats->structure_rebuilt stays 0 and
 * stats->t_symbolic is the host time of the in-place extension.  A factor outside the band (a far loop closure) or an
 * exhausted reserve triggers one ordinary rebuild (with a fresh reserve).  Defaults 384 / 64; reserve 0 disables. */
int fgo_isam2_reserve(fgo_ctx *ctx, int reserve_variables, int window);
/* The same growth reserve for g2o-semantics contexts: CGraphG2O::addNode adds key frames -- each matched against its
 * predecessor and the m_lookback_nodes before it (g2o/g2o_graph.cpp:159-239) -- between optimizeGraph() calls that come
 * every m_optimize_step key frames (g2o/test_g2o_graph.cpp:80-83).  g2o rebuilds its structure at every such call; here a
 * structure built in growth mode takes the new vertices into its reserve slots and the new edges (pairs inside the band
 * `window`) into its device arrays in place: fgo_optimize's stats->structure_rebuilt stays 0 and stats->t_symbolic is the
 * host time of the extension.  Growth mode switches itself on the first time a built structure has to be REBUILT because
 * vertices were added; fgo_set_growth(ctx, R, W) with R > 0 switches it on beforehand (W = 0: default band 16 >= 1 + m_lookback_nodes),
 * fgo_set_growth(ctx, 0, 0) switches it (and the automatic rule) off.  An edge outside the band (a far loop closure), a fixed new
 * vertex or an exhausted reserve costs one ordinary rebuild (with a fresh reserve).  The estimate does not depend on the mode
 * beyond rounding (another elimination order). */
int fgo_set_growth(fgo_ctx *ctx, int reserve_variables, int window);
/* ISAM2Params::wildfireThreshold analogue (gtsam/gtsam_graph.cpp:93-99 leaves GTSAM's default, 1e-3, in place).  0 (the
 * default here) = exact back-substitution of every variable at every update.  threshold > 0: below the top levels of the
 * elimination tree -- the re-factored root paths, always solved -- a task is solved again only if it was re-factored or an entry
 * of delta it depends on changed by >= threshold since the previous update; the others keep their delta.  Like GTSAM's, the
 * cut follows THIS elimination order, so the two approximations agree to the order of the threshold, not digit by digit.
 * The cut applies to an update only when (i) the update ran as a PARTIAL re-factorisation (stats->reserved[3] >= 0), (ii) the
 * schedule has a backward chain -- at least two panel levels at the top of the tree, single GPU -- and (iii) the previous
 * update left its solution behind; otherwise the call still returns FGO_OK and the back-substitution stays exact:
 * stats->reserved[4] of fgo_isam2_update says which of the two happened. */
int fgo_isam2_set_wildfire(fgo_ctx *ctx, double threshold);
/* delete mp_isam2; new ISAM2(params): forget theta and delta (the values stay).  Also leaves the incremental mode: the
 * growth reserve is dropped at the next use of the context and laid down again by the next fgo_isam2_update.  Batch
 * entry points (fgo_optimize_gtsam, marginals) called BETWEEN fgo_isam2_update calls keep the reserve (no structure
 * ping-pong in the reference's per-record flow); its cost is `reserve` identity columns coupled in a `window`-wide band. */
int fgo_isam2_reset(fgo_ctx *ctx);
/* ISAM2::getLinearizationPoint().at(key), ISAM2::getDelta()[key] (either output may be NULL) */
int fgo_isam2_get_state(fgo_ctx *ctx, int64_t id, double theta7[7], double delta6[6]);
/* NonlinearFactorGraph::error(values) = 0.5 * sum ||whitened r||^2 — CGraphGT::error, gtsam/gtsam_graph.cpp:173-176 */
double fgo_error(fgo_ctx *ctx);
/* Marginals(graph, values, Marginals::CHOLESKY).marginalCovariance(key) — gtsam/gtsam_graph.cpp:598-601: the 6x6
 *      (row-major, tangent order of the graph's semantics; 3-dof variables use the top-left 3x3) diagonal block of
 *      (J' Omega J)^-1 at the current estimate.  Works for both semantics. */
int fgo_marginal_cov(fgo_ctx *ctx, int64_t id, double *cov36);
/* Several blocks from ONE factorisation (the reference asks for many per Marginals object: gtsam/gtsam_graph.cpp:598-601,
 * :1357 + the commented-out association test :1413-1414): cov36 = n x 36 doubles.  The undamped factor stays resident in
 * HBM until the estimate or the structure changes, so consecutive calls do not re-factor either. */
int fgo_marginal_cov_many(fgo_ctx *ctx, int64_t n, const int64_t *ids, double *cov36);

/* ---- solve: ONE SparseOptimizer::optimize(max_iters) call as issued by
 *      CGraphG2O::optimizeGraph (g2o/g2o_graph.cpp:246-249).  Returns the number of LM iterations
 *      performed (>= 1), FGO_ESTATE if there is nothing to optimise, or another negative code. */
int fgo_optimize(fgo_ctx *ctx, int max_iters, fgo_stats *stats /* may be NULL */);
/* computeActiveErrors(); chi2()  — CGraphG2O::error, g2o/g2o_graph.cpp:254-258 (no 1/2).  NaN on error. */
double fgo_chi2(fgo_ctx *ctx);
/* per-iteration (chi2, lambda) of the last fgo_optimize call; returns the number written */
int fgo_trace(const fgo_ctx *ctx, double *chi2s, double *lambdas, int cap);

/* ---- building blocks exposed for parity tests and profiling (same device kernels the solve uses).
 * fgo_linearize: computeActiveErrors + buildSystem at the current estimate; optional outputs are the
 * dense (6*n_free)^2 row-major H and 6*n_free b in free-variable order = order in which the variables were added (small graphs
 * only: n_free <= 4096).  fgo_solve_step: one damped solve (H + lambda I) d = b, d returned in the same
 * order.  n_free counts the CALLER's free variables only (fgo_stats.n_free, *n_free_out): the phantom slots a context in
 * incremental mode keeps behind them (fgo_isam2_reserve) are internal and never appear in H_dense, b_dense or delta_out,
 * so buffers sized from the caller's own free-variable count are always large enough; landmarks a bundle-adjustment structure
 * eliminates analytically DO count (asking for the dense system switches the context to the generic form).  fgo_bench_phase: repeats one phase as an LM trial runs it (0 linearize, 1 factor sweep with
 * the forward solve fused in, 2 backward solve sweep) 'reps' times on the context's stream and returns the mean device ms
 * per repetition. */
int fgo_linearize(fgo_ctx *ctx, double *chi2_out, double *H_dense, double *b_dense, int64_t *n_free_out);
int fgo_solve_step(fgo_ctx *ctx, double lambda, double *delta_out);
int fgo_bench_phase(fgo_ctx *ctx, int phase, int reps, double *ms_out);
int fgo_get_stats(const fgo_ctx *ctx, fgo_stats *stats);       /* structure fields of the last build */

/* ---- synthetic pose graphs (SURVEY.md §8d "Manhattan-3D"); host-only, no device needed.
 * Lattice random walk, odometry + `lookback` look-back edges per pose as CGraphG2O::addNode builds
 * them (g2o/g2o_graph.cpp:196-205) + up to `n_loop` loop closures to earlier poses within 2 m.
 * Outputs must hold n_poses*7 / max_edges*{1,1,7,21} entries; returns the edge count (or <0). */
int64_t fgo_synth_manhattan3d(int64_t n_poses, int lookback, int n_loop, uint64_t seed, double sigma_t,
                              double sigma_q, double *poses_init7, double *poses_true7, int64_t *id_i,
                              int64_t *id_j, double *meas7, double *info_ut21, int64_t max_edges);

/* ---- multi-GPU: distributed factorisation by domain decomposition (north star: "the graph shards by
 * pose-block column across up to 8 GPUs with RCCL all-reduce ... on the off-diagonal Hessian contributions").
 * The reference's only solve site is single-threaded (g2o/g2o_graph.cpp:246-249).  Here every rank holds the whole graph
 * (host side) and calls the same entry points in the same order -- fgo_optimize* and fgo_chi2 become COLLECTIVE calls.
 * fgo_set_shard(rank, world) cuts the elimination tree into `world` groups of sub-trees ("domains": contiguous ranges of
 * block columns, one group per rank) plus their common ancestors (the "top": the upper nested-dissection separators).
 * A rank linearises only the factors of its domain, factors only its own block columns (with the forward solve fused),
 * and adds its updates into the top's blocks of L and entries of the right-hand side; ONE all-reduce per LM trial sums
 * those contributions -- exactly the Hessian blocks and updates that cross from a domain's columns into the separator
 * columns -- then every rank finishes the (small, latency-bound) top redundantly, back-substitutes through the top and
 * its own domain, updates its poses and re-linearises.  Scalars (chi2, the LM scale, lambda_0) are summed / maximised
 * over the ranks, so all ranks take identical accept / reject decisions.  At the end of an optimize call the ranks'
 * poses are gathered, so fgo_get_pose* answers with the whole estimate on every rank.
 * Transport: fgo_dist_init_rccl (RCCL on the context's stream: no host callback, no extra synchronisation), or a
 * host callback (fgo_set_allreduce: tests, torch.distributed) that must sum a device buffer over the ranks in place.
 * ISAM2 updates, marginal covariances and fgo_solve_step are single-GPU entry points (FGO_ESTATE when world > 1).
 * Errors: fgo_optimize* first agree on a status word, so a rank whose structure build failed (or that has nothing to
 * optimise) makes ALL ranks return an error instead of leaving them blocked in a collective; numerical failures inside
 * the LM loop are agreed through the scalar collective.  A HIP or transport error in the middle of a trial is fatal for
 * the communicator (the other ranks may block): destroy the contexts. */
typedef int (*fgo_allreduce_fn)(void *user, double *device_buffer, int64_t count);
int fgo_set_shard(fgo_ctx *ctx, int rank, int world);
int fgo_set_allreduce(fgo_ctx *ctx, fgo_allreduce_fn fn, void *user);
/* RCCL transport: rank 0 obtains a 128-byte id (ncclGetUniqueId), the host program broadcasts it to all ranks by any means,
 * every rank calls fgo_dist_init_rccl after fgo_set_shard (ncclCommInitRank; one GPU per rank).  librccl is loaded at
 * run time (FGO_RCCL_LIB overrides the name), so single-GPU deployments do not need it.  An id serves ONE communicator
 * (one rendezvous): a second context, or a context that is re-initialised, draws and broadcasts a new one. */
int fgo_dist_unique_id(void *id128);
int fgo_dist_init_rccl(fgo_ctx *ctx, const void *id128);
/* tests: the decomposition for a block graph (host only; group_out[v] = owning rank, `world` = top); one all-reduce of
 * host data through the context's transport */
int fgo_debug_partition(int n, int64_t n_pairs, const int *a, const int *b, int world, int *group_out);
int fgo_debug_allreduce(fgo_ctx *ctx, double *host_buffer, int64_t count);
/* contiguous shard [lo, hi) of n items for rank r of w (host-only helper, also used internally) */
int fgo_shard_range(int64_t n, int rank, int world, int64_t *lo, int64_t *hi);
/* profiling hook: the first `count` doubles of the device-side reduction scratch (FGO_TRI_PROF=1 makes single-panel
 * k_panel_tri launches leave shader-clock stamps of their phases there: tools/tri_prof.py) */
int fgo_debug_read_scratch(fgo_ctx *ctx, double *out, int64_t count);
/* debugging / tests: copy the current (partial or full) H blocks, b and chi2 to the host.
 * H: n_hblocks*36 doubles (fgo_stats.nnz_H_blocks), b: 6*n_free doubles.  FGO_ESTATE on a structure that carries the
 * growth reserve of the incremental mode. */
int fgo_debug_read_system(fgo_ctx *ctx, double *H, double *b, double *chi2);
/* tests: the REDUCED camera system of a structure whose Point3 landmarks are eliminated first (the Schur complement GTSAM's
 * multifrontal elimination forms for gtsam/gtsam_graph.cpp:370-448 graphs): S = H_cc - W (H_pp + lambda I)^-1 W^T,
 * g = b_c - W (H_pp + lambda I)^-1 b_p at the current estimate, dense row-major (6 n)^2 / 6 n with n = *n_out = the free
 * non-landmark variables in the order they were added (n <= 4096).  FGO_ESTATE if no landmarks are eliminated. */
int fgo_debug_read_reduced(fgo_ctx *ctx, double lambda, double *H_dense, double *b_dense, int64_t *n_out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FGO_H */
