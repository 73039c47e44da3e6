/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_se3.h header).  Shared problem state of orc_graph.c / orc_gtsam.c. */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "orc_api.h"
struct orc_problem {
  int N, E, nfree;
  double *poses, *backup;
  unsigned char *fixed;
  int *ei, *ej;
  double *meas, *info;
  int *hidx;              /* pose id -> hessian block index or -1 */
  /* GTSAM-semantics extension (orc_gtsam.c): per-edge kind, unary Pose3 priors, exponential-map retraction */
  int manifold;           /* 0 = g2o VertexSE3 oplus, 1 = GTSAM Pose3 Expmap chart */
  int *kind;              /* NULL or per edge: 0 = g2o EdgeSE3, 1 = BetweenFactor<Pose3>, 2 = OrientedPlane3Factor
                             (pose, plane; meas = z[4], info = 3x3 upper in info[0..5]),
                             3 = GenericProjectionFactor<Pose3,Point3,Cal3DS2> (pose, point; meas = uv, info[0] = 1/sigma^2) */
  int *vkind;             /* NULL or per variable: 0 = Pose3, 1 = OrientedPlane3 (n,d), 2 = Point3, 3 = Vector3, 4 = bias(6) */
  double calib[9];        /* Cal3DS2: fx fy s u0 v0 k1 k2 p1 p2 */
  double body_P_sensor[7];
  /* CombinedImuFactor(X_i, V_i, X_j, V_j, B_i, B_j): 6 variable ids, preintegrated payload, 15x15 information */
  int nimu;
  int *imu_ids;           /* 6 per factor */
  void *imu_pre;          /* orc_preint[nimu] (orc_imu.h) */
  double *imu_info;       /* 225 per factor, row-major, order theta p v ba bg */
  int *imu_blk;           /* 15 per factor: off-diagonal block of each variable pair (u < w), or -1 */
  double gravity[3];
  int nprior;
  int *pv;                /* prior -> pose id */
  double *pmean, *pinfo;  /* 7 / 21 per prior */
  /* block structure (built lazily) */
  int built;
  int nblk;               /* off-diagonal blocks (r < c in hessian index space) */
  int *blk_r, *blk_c;     /* sorted by (c, r) */
  int *edge_blk;          /* edge -> block index or -1 */
  double *Hd, *Ho, *b;    /* diag blocks [nfree*36], off-diag [nblk*36] (row-major, rows = blk_r) */
  int *perm, *iperm;      /* block AMD ordering */
  /* scalar CSC of the permuted upper triangle */
  int *Cp, *Ci;
  double *Cx;
  long long *colbase;     /* per permuted block column: base offset in Cx */
  int *colm;              /* per permuted block column: # off-diag blocks */
  int *blk_rank;          /* per block: rank inside its permuted block column */
  int *blk_pc;            /* per block: permuted block column */
  unsigned char *blk_tr;  /* per block: stored transposed in permuted space */
  orc_chol *chol;
  orc_sn *sn;             /* supernodal factor (built on first use with orc_set_solver(1)) */
  double *x;              /* solution (hessian index order, 6*nfree) */
  double *xp;             /* permuted work */
  double t_symbolic;
  /* trace */
  double tr_chi2[256], tr_lambda[256];
  int ntrace;
};

double orc_now_s(void);
void orc_build_structure(orc_problem *p);
double orc_linearize(orc_problem *p);
int orc_solve(orc_problem *p, double lambda, double *t_factor, double *t_solve);
void orc_apply_update(orc_problem *p);
/* factor / variable dispatch (orc_gtsam.c).  Residuals and Jacobians are padded to 6 rows / 6 columns. */
void orc_factor_eval(const orc_problem *p, int k, double e[6], double *Ji, double *Jj, double W[36]);
void orc_prior_dispatch(const orc_problem *p, int k, double e[6], double *J);
double orc_imu_chi2(const orc_problem *p);
double orc_imu_linearize(orc_problem *p);      /* adds the IMU factors to Hd / Ho / b, returns their chi2 */
int orc_var_dim(int vkind);
void orc_var_retract(int vkind, const double *x, const double *d, double *out);
#endif
