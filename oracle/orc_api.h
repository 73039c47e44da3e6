/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_se3.h header).  C API of the CPU restatement,
 * loaded through ctypes by tests/ and by the cpu_baseline leg of bench.py. */
#ifndef ORC_API_H
#define ORC_API_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- ordering / sparse Cholesky (stand-in for g2o::LinearSolverCSparse) ---- */
int orc_amd_order(int n, const int *xadj, const int *adj, int *perm);

typedef struct orc_chol orc_chol;
/* C = upper triangle (row <= col) of the already-permuted SPD matrix, CSC */
orc_chol *orc_chol_symbolic(int n, const int *Cp, const int *Ci);
int orc_chol_numeric(orc_chol *c, const int *Cp, const int *Ci, const double *Cx);
/* multi-threaded leg of the CPU baseline (OpenMP, elimination-tree sub-trees in parallel): same factor, bit for bit */
int orc_chol_numeric_mt(orc_chol *c, const int *Cp, const int *Ci, const double *Cx, int nthreads);
void orc_set_threads(int n);   /* threads used by orc_optimize's linearisation and factorisation (default 1) */
int orc_get_threads(void);
void orc_chol_solve(const orc_chol *c, double *x);
long long orc_chol_nnz(const orc_chol *c);
int orc_chol_etree_height(const orc_chol *c);
void orc_chol_free(orc_chol *c);
/* supernodal left-looking Cholesky on dense panels (orc_chol_sn.c): the second CPU leg ("what a tuned CPU solver would do") */
typedef struct orc_sn orc_sn;
orc_sn *orc_sn_symbolic(int n, const int *Cp, const int *Ci);
int orc_sn_numeric(orc_sn *c, const double *Cx, int nthreads);
void orc_sn_solve(const orc_sn *c, double *x);
long long orc_sn_nnz(const orc_sn *c);
int orc_sn_count(const orc_sn *c);
void orc_sn_free(orc_sn *c);
void orc_set_solver(int kind);  /* 0 = simplicial up-looking (g2o's LinearSolverCSparse class, default), 1 = supernodal */
int orc_get_solver(void);

/* ---- per-factor arithmetic ---- */
void orc_edge_se3_eval(const double *xi, const double *xj, const double *z, double *e, double *Ji,
                       double *Jj);
void orc_pose_oplus_eval(const double *x, const double *d, double *out);

/* ---- pose-graph problem with g2o semantics ---- */
typedef struct orc_problem orc_problem;
typedef struct {
  int iterations;        /* LM iterations performed in this optimize() call */
  int trials;            /* linear solves (accepted + rejected) */
  int terminated;        /* g2o 'Terminate' result seen */
  double chi2_initial, chi2_final, lambda_final;
  double t_symbolic, t_linearize, t_factor, t_solve, t_update, t_total;
  long long nnz_H_blocks, nnz_L_scalar;
} orc_stats;

orc_problem *orc_create(int n_poses, const double *poses7, const unsigned char *fixed, int n_edges,
                        const int *ei, const int *ej, const double *meas7, const double *info21);
void orc_free(orc_problem *p);
double orc_chi2(const orc_problem *p);                 /* sum e' Omega e, g2o_graph.cpp:254-258 */
int orc_optimize(orc_problem *p, int iterations, orc_stats *st);  /* ONE SparseOptimizer::optimize(n) */
void orc_get_poses(const orc_problem *p, double *poses7);
void orc_set_poses(orc_problem *p, const double *poses7);
/* dense H (nfree*6 square, row-major) and b for tiny graphs; free-variable order = ascending id */
int orc_dense_system(const orc_problem *p, double *H, double *b, int *n_free_out);
/* per-iteration chi2 / lambda trace of the last optimize() call (up to cap entries) */
int orc_trace(const orc_problem *p, double *chi2s, double *lambdas, int cap);
/* one undamped-or-damped linear step for tests: solves (H + lambda I) d = b, returns d (6*nfree) */
int orc_solve_step(orc_problem *p, double lambda, double *delta);

#ifdef __cplusplus
}
#endif
#endif
