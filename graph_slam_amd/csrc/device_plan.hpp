// Device-resident layout of one optimisation problem (passed by value to every kernel).
// All arrays live in HBM and are owned by the context.  Layout choices (DESIGN.md "Data layout"):
//   poses      AoS, padded to 8 doubles (64 B) so a gather is two 32-B vector loads
//   edges      one 256-byte record per edge (EDGE_REC doubles): [0..6] inverse measurement, [8..28] information (upper
//              triangle); a half-edge gather reads two whole 128-byte lines whichever side it comes from; edge_i/j[E]
//   H          36-double row-major blocks: [0, nb) diagonal blocks in elimination order, then the
//              off-diagonal blocks, already oriented (row = later-eliminated pose) for the factor
//   L          36-double row-major blocks in block-CSC order (diag first, ascending rows)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace fgo {

// Cal3DS2 intrinsics + body_P_sensor shared by all reprojection factors of a context
// (gtsam/gtsam_graph.cpp:373, 405-409)
struct CamCalib {
  double fx, fy, s, u0, v0, k1, k2, p1, p2;
  double bps[7];        // body_P_sensor as t(3) q(4)
  double ad[36];        // AdjointMap(body_P_sensor^-1) = d compose(X, B) / d X, row-major
};

// one CombinedImuFactor's payload in HBM: 288 doubles (2304 B)
struct ImuPayload {
  double dt;
  double dR[4];
  double dp[3], dv[3];
  double J_R_bg[9], J_p_ba[9], J_p_bg[9], J_v_ba[9], J_v_bg[9];
  double bhat[6];
  double info[225];      // row-major symmetric 15x15 information (inverse of preintMeasCov), order theta p v ba bg
  double pad;
};
static_assert(sizeof(ImuPayload) == 288 * sizeof(double), "ImuPayload layout");

// one 16/32-byte descriptor per panel / workgroup instead of chains of dependent index loads (each dependent load
// costs a microsecond of memory latency in kernels that only live for ten)
struct AccDesc { int t, a; long long o0, o1; };   // a target of the gather-form accumulate: block, where its value starts (>= 0: H block a, -1: zero, -2: its value in L), ops [o0, o1) left to the launch
struct LeafDesc { int task, c_begin, m, k0; long long base, obase; int nblk, nops; };   // a light sub-tree of a leaf level (k_chol_leaf): first column, its first block / op, counts
struct PanelDesc { int task, m, cols0, prow0, nrows, chunk0, nchunks, top; };   // cols0: first entry in task_cols; top: slot in ptop (-1: none)
struct RowChunk { int pn, m, s0, R6, prow0, cols0, top, task; };                // 16 scalar rows of the row kernel; top: slot in ptop
struct BwdChunk { int pn, m, row0, nrows; };                                    // <= PANEL_ROWS block rows (absolute row0)
struct ChainItem { int pn, need; };                                             // k_bwd_chain: panel + panels of the levels above it (all must be done first)

// panels (fgo_internal.hpp, Symbolic): descriptors of the supernode-like column paths at the top of the tree
struct PanelPlan {
  const PanelDesc *pdesc;
  const LeafDesc *leaf_desc, *leaf_lpt;   // [tasks]: descriptors of the leaf levels' tasks in task order (partial sweeps) / per level by descending work (full sweeps)
  const int *tri_order;           // [n_panels] launch order of the throughput triangle kernels within a level: by panel width (full sweeps)
  const RowChunk *rchunks;
  const BwdChunk *bchunks;
  const ChainItem *bchain;        // backward chain: the panels of the top levels, root level first (HostSchedule::bchain_*)
  unsigned *bchain_done;          // its progress counter (0 between launches)
  const int *task_panel, *panel_task;
  const int *ptri_blk;            // [n_panels][PM*PM]
  const int *prow_ptr, *prow_idx, *prow_blk;
  const int *pchunk_panel, *pchunk_row0, *pchunk_nrows, *panel_chunk0;
  const int64_t *row_mid;         // [nb]
  const int *fchunk_col;
  const int64_t *fchunk_e0;
  const int *pcol_fchunk0, *pcol_fchunkn;
  const int *rchunk_panel, *rchunk_s0;   // row-kernel chunks: 16 scalar rows of a panel's off-triangle rows
  const int *rchunk_src;          // [row chunks from rchunk_src0 on][16 scalar rows][PANEL_MAX]: prow_src of the chunk's rows (the right-hand-side row: its columns), -1 beyond:
  int rchunk_src0;                //   addressed by the CHUNK, so the row kernel of the NARROW levels requests it beside the chunk's descriptor (one dependent round trip less)
  const int *ptri_src, *prow_src; // like ptri_blk / prow_blk: >= 0 value in L block, <= -2 value in H block -2-x, -1 zero
  double *ptop;                   // [panels of panel levels][NJ (NJ+1)/2 tiles of 256] factored triangles as MFMA operand tiles (k_panel_tri)
  double *fpart;                  // [n_fchunks][6]   partial forward sums
  double *bpart;                  // [n_pchunks][PM][6] partial backward sums
};

struct RideItem;

constexpr int EDGE_REC = 32;    // doubles per edge record (256 B = two 128-byte lines)
// Linearisation hubs.  A variable with many half-edges would serialise its factor evaluations on the 4 lanes it normally
// gets; above DevPlan::hub_deg it is linearised by whole 256-thread workgroups instead, one per SLICE of HUB_SLICE
// half-edges (a plane seen from 50 000 keyframes: 25 workgroups, 8 evaluations per thread), the slices' partial sums
// combined in a fixed order by k_hub_combine*.  hub_deg is chosen per graph (fgo_api.cpp plan_hubs): the smallest of
// 64 .. 1024 that leaves at most HUB_MAX_VARS hub variables -- a few hundred planes seen from everywhere become hubs at
// 64; the 10 000 cameras of a bundle adjustment (~500 observations each, plenty of them to fill the chip) stay on the
// 4-lane path, which measured faster for them.
constexpr int HUB_DEG = 1024;
constexpr int HUB_SLICE = 512, HUB_MAX_SLICES = 64, HUB_MAX_VARS = 1024, HUB_PART = 42;

// ---- Bundle adjustment with the landmarks eliminated first (kernels_ba.hip; gtsam/gtsam_graph.cpp:370-448, 500-610 build
// such graphs).  A free Point3 variable that only carries reprojection factors (and unary priors) is not a column of the block
// system: per LM trial its 3x3 block (H_pp + lambda I) is factored on the spot, the reduced camera system
//   S = H_cc - sum_p Y_p Y_p^T,   Y_(c,p) = W_(c,p) L_pp^-T  (6x3 per observation),   rhs_c = b_c - sum_p Y_(c,p) y_p
// goes through the block-sparse Cholesky unchanged, and the landmarks follow by back-substitution.  Observations are numbered
// camera-major (by the camera's column, then by landmark), so a camera's Y blocks are contiguous.
struct BaPlan {
  int n_lm;                      // eliminated landmarks (0: mode off)
  const unsigned char *lm_mine;  // distributed mode: [n_lm] 1 = this rank eliminates the landmark (all of its observations are its
                                 // factors; the others' lists are empty here and their per-landmark work is skipped); NULL: all
  int64_t n_obs;                 // their observations
  int n_tgt, n_cam;              // blocks of S that receive landmark terms; camera columns with observations
  const int *lm_var;             // [n_lm] variable of a landmark (its virtual column is nb + index: b / x only)
  const int64_t *pt_ptr;         // [n_lm + 1] -> pt_obs
  const int *pt_obs;             // observations of a landmark, ascending camera column
  const double *pt_uvw;          // [n_obs][3] ... their pixel measurement and weight again, in THIS order (k_ba_linearize streams them)
  const int *pt_cam;             // [n_obs] ... and camera variable
  const int64_t *lp_ptr;         // [n_lm + 1] unary priors of a landmark, in the landmarks' order ...
  const double *lp_val;          // ... [9] each: mean x y z, information 00 01 02 11 12 22
  const double *obs_uvw;         // [n_obs][3] pixel measurement and weight (1 / sigma^2) of the observation
  const int *obs_cam;            // [n_obs] camera variable
  const int *obs_col;            // [n_obs] its column, -1: fixed camera (contributes to the landmark's own block only)
  const int *obs_lm;             // [n_obs] landmark index
  const int64_t *cam_ptr;        // [n_cam + 1] observation range of a camera column
  const int *cam_col;            // [n_cam]
  int64_t o_first;               // observations [0, o_first) belong to fixed cameras (no coupling block)
  int n_tgt_small;               // tgt_list: first the n_tgt_small blocks with short lists (one wave each), then the long ones
  const int *tgt_list;           // [n_tgt_list] the blocks k_ba_schur still takes one by one (cameras with more row blocks than k_ba_schur_cam holds)
  int n_tgt_list;
  // k_ba_schur_cam: the blocks of the reduced system grouped by COLUMN camera -- blocks [cam_t0[i], cam_t0[i + 1]) (and their
  // pair lists) belong to camera i of cam_col; cam_list = the cameras that kernel takes
  const int64_t *cam_t0;         // [n_cam + 1]
  const int *cam_list;           // [n_cam_list]
  int n_cam_list;
  const int *tgt_blk;            // [n_tgt] H block (row = the later column)
  const int64_t *tgt_ptr;        // [n_tgt + 1] -> ops
  const int *op_a, *op_b;        // observation of the row camera / of the column camera, one pair per shared landmark
  const int *op_lm;              // ... and that landmark
  double *pt_val;                // [n_lm][3] the landmarks' positions in their order: written by k_ba_linearize, read by k_ba_cameras (same linearisation)
  double *Hinv, *zp;             // per trial: [n_lm][6] (H_pp + lambda I)^-1 as h00 h01 h02 h11 h12 h22, [n_lm][3] its product with b_p
};

struct DevPlan {
  PanelPlan pp;
  // graph
  int64_t n_poses, n_edges;
  int64_t n_real;               // variables [n_real, n_poses) are unclaimed growth slots (identity diagonal, no update): n_poses when there are none
  int64_t edge_stride;          // capacity of the edge arrays in edges (>= n_edges: incremental mode keeps room for later factors)
  // 6-variable IMU factors: payload, variable ids, H slot of each of the 15 pairs.  This rank's factors are listed by COLOUR
  // (two factors of one colour share no variable: fgo_structure.cpp imu_colour); a launch per colour adds every factor's
  // blocks straight into H, without atomics and in a fixed order.
  int64_t n_imu;
  const ImuPayload *imu;
  const int *imu_ids;           // [6 n_imu]
  const int *imu_slot;          // [15 n_imu] (H block << 1 | transpose) or -1, pair order (0,1),(0,2)..(4,5)
  int64_t imu_fn;               // this rank's IMU factors: imu_list[0 .. imu_fn), sorted by colour
  const int *imu_list;
  int imu_ncolor;               // colours: imu_list[imu_color_ptr_h[c] .. imu_color_ptr_h[c + 1])
  const int *imu_color_ptr_h;   // HOST pointer (read by launch_linearize_gtsam only)
  double *imu_stash;            // [n_imu][150] scratch: r and the fifteen 3x3 Jacobian pieces of a factor (k_imu_eval)
  double gravity[3];
  const int *var_kind;         // [n_poses] 0 pose, 1 plane, 2 point, 3 vec3, 4 bias   (NULL in g2o mode)
  const int *edge_kind;         // [E] 0 g2o EdgeSE3, 1 between, 2 plane factor, 3 reprojection (NULL in g2o mode)
  CamCalib cam;
  int nb;                       // free poses = block columns
  int64_t n_hblocks;            // nb diagonal + unique off-diagonal H blocks
  // multi-GPU shard mode: this rank's kernels see only its factors, so the sums are partial
  int lin_priors;               // 1: this rank linearises the unary priors and adds the padding identity (rank 0)
  int zero_offdiag;             // 1: clear the off-diagonal H area before linearising (blocks without a local writer)
  const int *pose_col;          // [n_poses] elimination position of a pose, -1 if fixed
  const int *edge_i, *edge_j;   // [E] internal pose indices
  const double *ainv;           // [E][EDGE_REC]: record of edge e at ainv + EDGE_REC e; [0..6] Z^-1 as t(3) q(4) (raw payload for plane / reprojection factors)
  const double *info;           // = ainv + 8: [8..28] of the record, upper triangle, row-major
  const int *edge_slot;         // [E] (H block index << 1 | transpose) or -1 (no off-diagonal block / duplicate)
  const int *hub_list;          // [n_hubs] hub ENTRIES: the variable of entry q (one workgroup per entry) ...
  const int *hub_slice;         // [n_hubs] ... and (slice | n_slices << 16) of that variable
  int n_hubs;
  int hub_deg;                  // variables with more half-edges than this are hubs
  double *hub_part;             // [n_hubs][HUB_PART] partial sums of the entries of multi-slice hubs
  const int *hubm;              // [n_hub_multi][3]: variable, first entry, slices of the hubs with more than one slice
  int n_hub_multi;
  const int64_t *he_ptr;        // [n_poses+1]
  const int *he;                // [2E] (edge << 1) | side
  int64_t n_dup_groups;
  const int64_t *dup_ptr;
  const int64_t *dup_edges;
  const int *dup_slot;
  // unary Pose3 priors (GTSAM path): CSR per pose + SoA payload
  int64_t n_priors;
  const int64_t *prior_ptr;     // [n_poses+1]
  const int *prior_pose;        // [n_priors]
  const double *prior_minv;     // [7][n_priors]  inverse of the prior mean
  const double *prior_info;     // [21][n_priors]
  // factor structure
  const int64_t *colptr;        // [nb+1]
  const int *rowidx;            // [nnzL]
  const int *asrc;              // [nnzL] H block feeding this L block, -1 = fill-in
  int prof_tri;                 // FGO_TRI_PROF=1: single-panel k_panel_tri launches record shader-clock stamps of their phases in `partial`
  int zero_blk;                 // index of an all-zero block at the end of L (padding for batched updates)
  const int64_t *op_ptr;        // [nnzL+1]
  const int64_t *op_mid;        // [nnzL]
  const int *op_a, *op_b;       // [nops]
  const int *acc_targets;
  // column-group accumulate (k_chol_acc2; Symbolic::g2_*): groups of ACC2_G targets of one column
  const AccDesc *acc_desc;      // parallel to acc_targets: everything the head of k_chol_acc used to chase through acc_start / op_ptr / op_mid / asrc / top_ext0
  const int *g2_tgt;            // [groups][ACC2_G] target block or -1
  const int64_t *g2_ptr;        // [groups+1] -> entries
  const int *g2_b;              // [entries] block (k, j): the B operand, the same for the whole group
  const int *g2_a;              // [entries][ACC2_G] block (i_g, j) or the zero block
  // ---- partial re-factorisation (incremental updates, fgo_isam2_update): only the tasks flagged in task_dirty are run, the
  // blocks of L and the entries of y of every other column keep the values of the previous factorisation.  NULL = all.
  const unsigned char *task_dirty;   // [ntask]
  const int *acc_task;          // [n_acc] task of every accumulate target (parallel to acc_targets)
  const int *g2_task;           // [groups] task of every column-group of k_chol_acc2
  const int *tcol_task;         // [nb] task of every entry of task_cols
  BaPlan ba;                    // landmark elimination (n_lm == 0: off)
  // forward-solve work items of the accumulate launches (kernels.hip fwd_role): entry of task_cols + chunk of its row (-1: whole
  // row); per entry of task_cols: first chunk / chunks of the row (when it is split)
  const int *fwg_ci, *fwg_ch;
  const int *fwd_f0, *fwd_fn;
  const int *fsplit_ci;         // entries of task_cols whose rows are split, grouped by level (k_fwd_combine)
  // riders (Symbolic::ride_items / acc_start; NULL: none)
  const RideItem *ride_items;
  int ride_xcd;                 // 1: rider workgroups take XCD-contiguous ranges of the items (FGO_RIDE_XCD)
  const int64_t *acc_start;     // [n_acc] parallel to acc_targets
  const int64_t *rowptr;        // [nb+1]
  const int *row_blk, *row_col;
  const int *task_ptr, *task_cols;
  // scratch for two-pass reductions
  double *partial;
  // ---- multi-GPU domain decomposition (fgo_set_shard; everything below is inert when dist == 0)
  // columns >= top_col0 / blocks >= top_blk0 form the "top" (the common ancestors of all domains): their values are
  // assembled by the ranks together -- each rank writes  H_partial - sum(updates sourced from ITS domain)  into the tail
  // of L (k_dist_acc) and  b_partial - sum(L_kj y_j, j in its domain)  into the tail of x (k_dist_rhs); one all-reduce
  // later every rank continues on the top with the remaining (top-sourced) updates.
  int dist;                     // 1: distributed factorisation
  int top_col0;                 // first top column (= nb when not distributed)
  int64_t top_blk0;             // first block of L in a top column (= nnzL when not distributed)
  const int64_t *top_ext0;      // [nnzL - top_blk0] first top-sourced op of a top block (its domain-sourced ops come before)
  const int64_t *own_op0, *own_op1;   // [nnzL - top_blk0] ops of a top block sourced from THIS rank's domain
  const int64_t *top_row0;      // [nb - top_col0] first top-sourced entry of a top column's row list
  const int64_t *own_row0, *own_row1; // [nb - top_col0] row-list entries of a top column that lie in THIS rank's domain
  const unsigned char *var_mine;      // [n_poses] this rank adds the padding identity / lambda-free unary terms of the variable (NULL: all)
  int lambda_rank;              // 1: this rank adds lambda to the diagonal blocks of the top (rank 0)
};

// level structure kept on the host to drive the launches
struct HostSchedule {
  int cus = 256;                    // compute units of the context's device: the launch thresholds are multiples of it
  int n_levels = 0;
  int world = 1, rank = 0;          // multi-GPU: a level is a segment (dependency level, group); seg_group[l] == world is the top
  std::vector<int> seg_group;
  int64_t n_top_blocks = 0;         // blocks of L in top columns
  int n_top_cols = 0;
  std::vector<int> level_ptr;
  std::vector<int64_t> acc_ptr, acc_mid;   // level l: targets [acc_ptr[l], acc_mid[l]) short lists, [acc_mid[l], acc_ptr[l+1]) long
  std::vector<int> ride_ptr;               // level l: rider items [ride_ptr[2l], ride_ptr[2l+1]) carried by its k_panel_tri launch, [2l+1, 2l+2) by its k_panel_rows launch (empty: none)
  std::vector<int64_t> g2_lvl;             // level l: groups [g2_lvl[l], g2_lvl[l+1]) of the column-group accumulate (empty: gather form)
  std::vector<int> level_maxcol;   // largest column (blocks) among the level's tasks
  std::vector<int> level_maxrow;   // longest row list among the level's columns
  std::vector<int> level_maxtaskcols;   // most columns in one task of the level
  std::vector<char> level_leaf;         // level runs k_chol_leaf
  std::vector<int> level_leaf_maxblk, level_leaf_maxops;
  std::vector<char> level_panel;   // level consists of panels only -> panel kernels
  int rows_byc_level = 1 << 30;    // first level whose row launches use the by-chunk code table (PanelPlan::rchunk_src)
  std::vector<int> level_pn0;      // first panel id of a panel level (ids are consecutive within the level)
  std::vector<int> level_col_ptr;  // columns of level l = task_cols[level_col_ptr[l] .. level_col_ptr[l+1])
  std::vector<int> fwg_ptr;        // forward-solve work items of level l = [fwg_ptr[l], fwg_ptr[l+1])
  std::vector<int> fsplit_ptr;     // split rows of level l = fsplit_ci[fsplit_ptr[l] .. fsplit_ptr[l+1])
  std::vector<int> pchunk_ptr, fchunk_ptr, rchunk_ptr;   // per level: row chunks / forward-solve chunks / row-kernel chunks
  int bchain_low = -1, bchain_n = 0;   // backward chain (k_bwd_chain): levels [bchain_low, n_levels) in ONE launch of bchain_n workgroups; -1: none
};

void launch_linearize(const DevPlan &P, const double *poses, double *Hblk, double *bvec, double *scalar_out, hipStream_t s);
void launch_chi2(const DevPlan &P, const double *poses, double *scalar_out, hipStream_t s);
void launch_maxdiag(const DevPlan &P, const double *Hblk, double *scalar_out, hipStream_t s);
void launch_update(const DevPlan &P, const double *poses, double *cand, const double *x, const double *b,
                   const double *lambda_p, double *scalar_out, hipStream_t s);
// phase (multi-GPU only): PHASE_ALL = single GPU; PHASE_DOMAIN = this rank's segments, then its contributions into the
// tail of L / x (k_dist_acc, k_dist_rhs); PHASE_TOP = the top segments (after the collective)
enum { PHASE_ALL = 0, PHASE_DOMAIN = 1, PHASE_TOP = 2 };
// A partial sweep (ISAM2 update: a few dirty tasks on one or two root paths) launches, per level, only the index ranges that cover
// its dirty tasks -- the lists of a level are in task order (checked when the tables are built) -- instead of full grids whose
// workgroups look their flag up and leave: t_lo / t_hi per level (t_hi < t_lo: nothing dirty there), per task the [first, end) of its
// short / long accumulate targets, column groups and row chunks.  The kernels and their template choices are those of the full
// sweep, so the result is bit-identical.
struct PartialSweep {
  const int *t_lo, *t_hi;
  const int64_t *s0, *s1, *l0, *l1;
  const int *g0, *g1, *c0, *c1;
  const int *task_ptr;
};
void launch_factor(const DevPlan &P, const HostSchedule &H, const double *Hblk, double *Lv, const double *lambda_p,
                   int *fail_flag, hipStream_t s, const double *b = nullptr, double *x = nullptr, int phase = PHASE_ALL, const PartialSweep *ps = nullptr,
                   const double *b_full = nullptr);   // (b_full: distributed landmark elimination, see k_dist_rhs)
// wildfire back-substitution (kernels.hip k_wild_*): device arrays per task (run, dirty) / per column (chg), the previous solution
struct Wildfire { unsigned char *run, *chg; const unsigned char *dirty; const double *xprev; double thr; };
void launch_solve(const DevPlan &P, const HostSchedule &H, const double *Lv, const double *b, double *x, hipStream_t s,
                  bool fwd_done = false, int phase = PHASE_ALL, const Wildfire *wf = nullptr);
void launch_mix_rhs(const DevPlan &P, const double *b, const double *ysaved, double *x, const unsigned char *col_dirty, hipStream_t s);
void launch_copy_vec(const double *src, double *dst, int64_t n, hipStream_t s);
void launch_mask_poses(const DevPlan &P, const double *poses, double *out, const int *pose_group, int rank, int world, hipStream_t s);
int linearize_blocks(const DevPlan &P);
void launch_reduce(const double *partial, int64_t n, double *out, int mode, hipStream_t s);
void prepare_device_kernels();
void launch_zero(double *p, int64_t n, hipStream_t s);      // zero fill as a kernel node (capturable without memset nodes)
void launch_zero_flag(int *p, hipStream_t s);
void launch_pack_scalars(double *scal, const int *fail, hipStream_t s, double *host = nullptr);   // host: pinned host memory that also receives [4..6]
void launch_fetch_scalar(double *dst, const double *src_host, hipStream_t s);                     // *dst = *src_host (pinned host memory), as a kernel node
// incremental mode: n new edges staged as [n][28] (7 payload + 21 information) -> SoA arrays at positions e0 .. e0+n, stride E_cap
void launch_scatter_edges(const double *stage, int64_t n, int64_t e0, double *rec, hipStream_t s);
// GTSAM-semantics factors (kernels_gtsam.hip)
// (ba_W / ba_Hpp / ba_bp: linearisation of the eliminated landmarks -- [n_obs][18], [n_lm][6], [n_lm][3] -- when P.ba is on)
bool linearize_gtsam_maskable(const DevPlan &P);   // binary factors and priors only: the masked form below applies
void launch_linearize_gtsam_masked(const DevPlan &P, const double *poses, double *Hblk, double *bvec, double *scalar_out, hipStream_t s,
                                   const unsigned char *mask, double *chi_var);
void launch_linearize_gtsam(const DevPlan &P, const double *poses, double *Hblk, double *bvec, double *scalar_out, hipStream_t s,
                            double *ba_W = nullptr, double *ba_Hpp = nullptr, double *ba_bp = nullptr);
// landmark elimination (kernels_ba.hip)
int ba_linearize_blocks(const DevPlan &P);
void launch_ba_linearize(const DevPlan &P, const double *vals, double *W, double *Hpp, double *bp, double *Hblk, double *bvec, double *chi_partial, hipStream_t s);
// per trial: factor the landmark blocks with lambda, Y / y_p, then Hred = H - sum Y Y^T and bred = b - sum Y y_p (reduced part)
void launch_ba_reduce(const DevPlan &P, const double *W, const double *Hpp, const double *bp, const double *H, const double *b,
                      double *Hred, double *bred, const double *lambda_p, int *fail_flag, hipStream_t s);
// after the reduced solve: x of the landmarks (virtual columns nb + index)
void launch_ba_back(const DevPlan &P, const double *W, const double *bp, double *x, hipStream_t s);
void launch_chi2_gtsam(const DevPlan &P, const double *poses, double *scalar_out, hipStream_t s);
void launch_update_gtsam(const DevPlan &P, const double *poses, double *cand, const double *x, const double *b,
                         const double *lambda_p, double *scalar_out, hipStream_t s);
void launch_isam2_relin(const DevPlan &P, double *theta, double *delta, double thr, double *count_out, hipStream_t s, unsigned char *moved_out = nullptr);
void launch_isam2_estimate(const DevPlan &P, const double *theta, const double *x, double *delta, double *est, hipStream_t s,
                           unsigned char *moved_next = nullptr, double thr_next = 0);

}  // namespace fgo
