// Internal host-side context of libfgo (shared by the fgo_*.cpp translation units; not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/fgo.h"
#include "device_plan.hpp"
#include "fgo_internal.hpp"
#include "rccl_min.hpp"

namespace fgo {

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0, cap = 0;
  ~DevBuf() { release(); }
  void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; cap = 0; } }
  // contents are NOT preserved.  A buffer that has to grow gets 25 % headroom: graphs that grow by a few variables per
  // update (fgo_isam2_update after every record) then rebuild their structure without a round of hipFree / hipMalloc
  hipError_t alloc(size_t count) {
    if (p && count <= cap) { n = count; return hipSuccess; }
    const bool regrow = p != nullptr;
    release();
    cap = (count ? count : 1) + (regrow ? count / 4 : 0);
    const hipError_t e = hipMalloc((void **)&p, sizeof(T) * cap);
    if (e != hipSuccess) { p = nullptr; cap = 0; return e; }
    n = count;
    return hipSuccess;
  }
  template <class A>
  hipError_t upload(const std::vector<T, A> &h, hipStream_t s) {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess) return e;
    if (h.empty()) return hipSuccess;
    return hipMemcpyAsync(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, s);
  }
  void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
};

}  // namespace fgo

struct fgo_ctx {
  fgo_config cfg{};
  std::string err;
  // ---- host graph store
  std::unordered_map<int64_t, int> id2idx;
  std::vector<int64_t> ids;
  std::vector<double> poses;        // 7 per pose
  std::vector<unsigned char> fixed;
  std::vector<int> ei, ej;
  std::vector<double> meas, info;   // 7 / 21 per edge
  std::vector<int> torder;
  std::vector<int> prior_v;         // unary Pose3 priors (GTSAM path)
  std::vector<double> prior_mean, prior_info;
  bool gtsam_mode = false;          // decided at build(): GTSAM-semantics factors + exponential-map retraction
  bool structure_dirty = true;      // vertices / edges added since the last build
  bool host_poses_newer = true;     // host copy must be uploaded before the next device use
  bool dev_poses_newer = false;     // device copy must be downloaded before the next host read
  bool lin_valid = false;           // H/b/chi2 on the device match the current device poses
  // ---- device
  hipStream_t stream = nullptr;
  bool use_graph = true;
  fgo::Symbolic S;
  fgo::HostSchedule sched;
  fgo::DevPlan plan{};
  int64_t n_offdiag = 0;
  fgo::DevBuf<int> d_pose_col, d_edge_i, d_edge_j, d_edge_slot, d_he, d_dup_slot, d_rowidx, d_asrc, d_op_a, d_op_b,
      d_acc_targets, d_row_blk, d_row_col, d_task_ptr, d_task_cols, d_fail;
  fgo::DevBuf<int64_t> d_he_ptr, d_dup_ptr, d_dup_edges, d_colptr, d_op_ptr, d_op_mid, d_rowptr, d_g2_ptr;
  fgo::DevBuf<int> d_g2_tgt, d_g2_b, d_g2_a;
  fgo::DevBuf<fgo::RideItem> d_ride_items;
  fgo::DevBuf<int> d_fwg_ci, d_fwg_ch, d_fwd_f0, d_fwd_fn, d_fsplit_ci;
  fgo::DevBuf<int64_t> d_acc_start;
  fgo::DevBuf<double> d_ainv, d_partial, d_poses[2], d_H[2], d_b[2], d_x, d_L, d_scal;
  // bundle adjustment with the landmarks eliminated first (device_plan.hpp "BaPlan", kernels_ba.hip)
  struct BaSchur {
    bool on = false;
    int n_lm = 0;
    fgo::DevBuf<unsigned char> d_lm_mine;
    fgo::DevBuf<int> d_lm_var, d_pt_obs, d_tgt_list, d_obs_cam, d_obs_col, d_obs_lm, d_cam_col, d_tgt_blk, d_op_a, d_op_b, d_op_lm, d_pt_cam;
    fgo::DevBuf<int64_t> d_pt_ptr, d_cam_ptr, d_tgt_ptr, d_lp_ptr, d_cam_t0;
    fgo::DevBuf<int> d_cam_list;
    fgo::DevBuf<double> d_obs_uvw, d_pt_uvw, d_lp_val, d_W[2], d_Hpp[2], d_bp[2], d_Hinv, d_zp, d_pt_val, d_Hred, d_bred;
  } ba;
  bool ba_disable = false;          // a request the eliminated form cannot serve (marginal of a landmark) switched it off for this context
  fgo::DevBuf<int> d_task_panel, d_panel_task, d_ptri_blk, d_prow_ptr, d_prow_idx, d_prow_blk, d_pchunk_panel, d_pchunk_row0,
      d_pchunk_nrows, d_panel_chunk0, d_fchunk_col, d_pcol_fchunk0, d_pcol_fchunkn;
  fgo::DevBuf<int64_t> d_row_mid, d_fchunk_e0;
  fgo::DevBuf<double> d_fpart, d_bpart, d_ptop, d_imu_stash;
  fgo::DevBuf<int> d_rchunk_panel, d_rchunk_s0, d_ptri_src, d_prow_src;
  fgo::DevBuf<int> d_hub_list, d_hub_slice, d_hubm;
  fgo::DevBuf<double> d_hub_part;
  fgo::DevBuf<fgo::PanelDesc> d_pdesc;
  fgo::DevBuf<int> d_tri_order, d_rchunk_src;
  fgo::DevBuf<fgo::LeafDesc> d_leaf_desc, d_leaf_lpt;
  fgo::DevBuf<fgo::AccDesc> d_acc_desc;
  fgo::DevBuf<fgo::RowChunk> d_rchunks;
  fgo::DevBuf<fgo::BwdChunk> d_bchunks;
  fgo::DevBuf<fgo::ChainItem> d_bchain;
  fgo::DevBuf<unsigned> d_bchain_done;
  fgo::DevBuf<int64_t> d_prior_ptr;
  fgo::DevBuf<int> d_prior_pose, d_var_kind, d_edge_kind;
  std::vector<int> var_kind;        // per variable: 0 pose, 1 plane, 2 point, 3 vec3, 4 bias (factors_device.hpp)
  fgo::CamCalib cam{};                   // Cal3DS2 + body_P_sensor for the reprojection factors
  bool cam_set = false;
  std::vector<int> imu_ids;         // 6 internal variable indices per CombinedImuFactor
  std::vector<fgo::ImuPayload> imu_payload;
  double gravity[3] = {0.0, 0.0, 9.71};   // MakeSharedD(9.71): gtsam/imu_base.cpp:258-263
  int shard_rank = 0, shard_world = 1;    // multi-GPU: this context owns domain `rank` of `world`
  fgo_allreduce_fn ar_fn = nullptr;       // host-callback transport of the collectives (tests, torch.distributed)
  void *ar_user = nullptr;
  ncclComm_t rccl = nullptr;              // RCCL transport: collectives enqueued on the context's stream (fgo_dist_init_rccl)
  std::vector<int> pose_group;            // per variable: owning rank, world = top, -1 = fixed
  fgo::DevBuf<int> d_pose_group, d_imu_list;
  fgo::DevBuf<int64_t> d_top_ext0, d_own_op0, d_own_op1, d_top_row0, d_own_row0, d_own_row1;
  fgo::DevBuf<unsigned char> d_var_mine;
  fgo::DevBuf<double> d_status;           // [1] status word the ranks agree on at the start of a collective entry point
  fgo::DevBuf<double> d_gather;                // [8 N] masked poses (end-of-optimize gather) / [world] scalar exchange
  hipGraphExec_t dist_graph[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [buffer parity][0: domain phase, 1: top phase]
  double xgmi_bytes = 0;                  // bytes this rank handed to the collectives since the last fgo_optimize* call began
  fgo::DevBuf<fgo::ImuPayload> d_imu;
  fgo::DevBuf<int> d_imu_ids, d_imu_slot;
  fgo::DevBuf<double> d_prior_minv, d_prior_info;
  int cur = 0;                      // which of the double buffers holds the current estimate
  bool cov_factor_valid = false;    // d_L holds the undamped factor of the current linearisation (marginal covariances)
  std::vector<int> h_pose_col;      // host copy of pose_col (marginal covariances)
  // ---- ISAM2 state (fgo_isam2_update): linearisation point and linear solution per variable, variable order
  fgo::DevBuf<double> d_theta, d_delta;  // 8 / 6 doubles per variable
  // partial re-factorisation: d_L / d_y hold the factor and forward solution of the previous ISAM2 step (isam_L_valid);
  // an update re-runs only the tasks on the paths from the affected variables to the roots (task_dirty)
  bool isam_L_valid = false;
  int64_t isam_E_seen = 0, isam_NI_seen = 0, isam_NP_seen = 0;   // factors the previous step already knew
  std::vector<int> col_task;        // [nb] task of every column (host)
  fgo::DevBuf<double> d_y;
  fgo::DevBuf<unsigned char> d_moved, d_task_dirty, d_col_dirty, d_moved_next;
  unsigned char *h_flags = nullptr;  // pinned staging: [NX] moved | [ntask] task_dirty | [nb] col_dirty | [NX] moved_next | [NX] affected
  // fgo_isam2_update decides at its END which variables the next call will relinearise (delta and the threshold are known
  // then) and brings the flags to the host with its final synchronisation: the next call starts without a mid-stream one
  // masked re-linearisation (kernels_gtsam.hip): H / b of the side buffers and the per-variable chi2 of the previous update
  // are still in place, so only the variables whose factors changed are gathered again
  fgo::DevBuf<unsigned char> d_lin_mask;
  fgo::DevBuf<double> d_chi_var;
  bool isam_H_valid = false;
  std::vector<int> isam_set_tasks, isam_set_cols;      // entries of the pinned flag arrays set by the previous update (cleared sparsely)
  std::vector<int64_t> isam_set_aff;
  // partial sweeps launch index ranges instead of full grids (device_plan.hpp PartialSweep): per task [first, end) of its items in
  // the level-wise lists, the level of every task; tk_ok: every list was found in task order when the tables were built
  std::vector<int64_t> tk_s0, tk_s1, tk_l0, tk_l1;
  std::vector<int> tk_g0, tk_g1, tk_c0, tk_c1, task_level, lvl_lo, lvl_hi;
  bool tk_ok = false;
  // wildfire back-substitution (fgo_isam2_set_wildfire): previous solution in column order, per-task / per-column flags
  double wild_thr = 0;
  bool wild_valid = false;                           // d_xprev holds the solution of the previous update on THIS structure
  fgo::DevBuf<double> d_xprev;
  fgo::DevBuf<unsigned char> d_bwd_run, d_chg;
  bool isam_moved_valid = false;
  double isam_moved_thr = -1;
  int64_t isam_moved_nx = 0;
  size_t h_flags_cap = 0;
  fgo::DevBuf<int> d_acc_task, d_g2_task, d_tcol_task;
  int64_t isam_n = 0;               // variables the state covers (variables added later start at their initial value, delta 0)
  // ---- incremental mode (fgo_isam2_update on a growing graph): the structure is built for the graph PLUS a reserve of
  // phantom variables, each coupled to the `window` variables before it (DESIGN.md "Incremental updates").  New variables
  // claim phantom slots and new factors whose variable pairs already exist in the structure are appended in place
  // (refresh_factors) -- no ordering, no symbolic factorisation, no re-upload of the index lists.
  bool isam_incremental = false;
  // g2o-semantics graphs grow the same way (CGraphG2O::addNode between optimizeGraph() calls, g2o/g2o_graph.cpp:159-239: a new
  // vertex is matched against its predecessor and the m_lookback_nodes before it): once a built structure had to be rebuilt
  // because vertices were added -- or fgo_set_growth asked for it -- the next build lays the same reserve down
  bool grow_incremental = false;
  int grow_auto = 1;                // 0: never switch growth mode on by itself (fgo_set_growth(ctx, 0, 0))
  int64_t built_N = -1;             // variables the current structure was built for (-1: none yet)
  int n_phantom = 0;                // phantom variables of the current structure: the LAST n_phantom free (hessian) indices; never reported to callers
  int isam_reserve = -1, isam_window = -1;      // -1: defaults (FGO_ISAM_RESERVE / FGO_ISAM_WINDOW or 384 / 64); reserve 0 disables
  struct Incr {
    int64_t NX = 0, N_done = 0, E_done = 0, NI_done = 0, NP_done = 0, E_cap = 0, NI_cap = 0;
    int nb = 0;
    std::vector<int> hidx, pose_col;              // [NX]
    std::vector<uint64_t> ukey;                   // sorted (a << 32 | b), a < b hessian indices: the structure's off-diagonal pairs
    std::vector<int> edge_h, edge_slot;           // per edge: pair index (-1: none), H slot (-1: none / duplicate group)
    std::vector<int> pair_nbin, pair_first;       // per pair: binary factors on it, the first of them
    std::map<int, std::vector<int64_t>> dups;     // pairs carrying more than one binary factor
    std::vector<int64_t> he_ptr, imu_inc_ptr;     // [NX+1] incidence CSRs, kept on the host so that new factors are INSERTED
    std::vector<int> he, imu_inc;                 // (they attach to the newest variables: short suffix to move and to upload)
    int hub_deg = fgo::HUB_DEG;                        // the graph's hub limit (plan_hubs), kept while the structure is extended in place
    size_t hub_cap = 0;                           // hub entries the scratch buffers (d_partial, d_hub_part) were sized for
    bool valid = false;
  } inc;
  // IMU factors of this rank by colour (imu_colour_add / imu_colour_lists in fgo_structure.cpp): factors of one colour share no variable
  std::vector<uint64_t> imu_var_mask;           // [NX] colours 0 .. 63 taken at a variable
  std::vector<int> imu_flist, imu_fcolor;       // this rank's factors in input order, and their colours (>= 64: a colour of its own)
  std::vector<int> imu_color_ptr;               // [colours + 1] into the colour-sorted list on the device (d_imu_list)
  int imu_extra = 0;
  fgo::DevBuf<double> d_stage;
  int64_t n_priors_dev = 0;
  hipGraphExec_t trial_graph[2] = {nullptr, nullptr};
  hipEvent_t ev[6] = {};
  double *h_scal = nullptr;         // pinned: [0] chi2 cur, [1] scale, [2] maxdiag, [3] lambda, [4] chi2 cand
  int *h_fail = nullptr;
  double chi_cur = 0;
  // ---- results
  fgo_stats last{};
  std::vector<double> tr_chi2, tr_lambda;
};

namespace fgo {

inline int fail(fgo_ctx *c, int code, const std::string &msg) {
  if (c) c->err = msg;
  return code;
}
#define HIPCHK(c, call)                                                                         \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(c, e_ == hipErrorOutOfMemory ? FGO_ENOMEM : FGO_ENODEV,                       \
                  std::string(#call) + ": " + hipGetErrorString(e_));                           \
  } while (0)

// Nothing may unwind through the extern "C" boundary into a C / ctypes / ROS caller (include/fgo.h: every entry point
// returns a code): the entry points are function-try-blocks.
#define FGO_CATCH_INT(c)                                                                              \
  catch (const std::bad_alloc &) { return fgo::fail(c, FGO_ENOMEM, "out of host memory"); }               \
  catch (const std::exception &e) { return fgo::fail(c, FGO_EINVAL, std::string("exception: ") + e.what()); } \
  catch (...) { return fgo::fail(c, FGO_EINVAL, "unknown exception"); }
#define FGO_CATCH_NAN(c)                                                                              \
  catch (const std::exception &e) { if (c) c->err = std::string("exception: ") + e.what(); return std::numeric_limits<double>::quiet_NaN(); } \
  catch (...) { if (c) c->err = "unknown exception"; return std::numeric_limits<double>::quiet_NaN(); }

double now_s();
extern std::string g_create_error;
void destroy_graphs(fgo_ctx *c);
void pose_inv7(const double *a, double *o);
int upload_poses(fgo_ctx *c);
int download_poses(fgo_ctx *c);
int ensure_ready(fgo_ctx *c);
// fgo_structure.cpp
int build(fgo_ctx *c);
int refresh_factors(fgo_ctx *c);
// fgo_dist.cpp
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;      // (optional: several all-reduces of one trial as ONE launch)
  ncclResult_t (*GroupEnd)() = nullptr;
};
RcclApi *rccl_api();
int dist_allreduce(fgo_ctx *c, double *buf, int64_t n);
int dist_allreduce2(fgo_ctx *c, double *buf_a, int64_t na, double *buf_b, int64_t nb);   // two sums, one launch on RCCL
int dist_sum_scalars(fgo_ctx *c, int slot, int n);
int dist_max_scalar(fgo_ctx *c, int slot);
int dist_gather_poses(fgo_ctx *c);
int dist_agree(fgo_ctx *c, int rc);
int run_trial_dist(fgo_ctx *c, double lambda, double *chi_cand, double *scale, int *failed, fgo_stats *st);
// fgo_lm.cpp
int run_trial(fgo_ctx *c, double lambda, double *chi_cand, double *scale, int *failed, fgo_stats *st);
// linearise / factor / solve on the context's buffers `buf` (0 / 1: current / candidate), including the landmark side when
// landmarks are eliminated (fgo_ctx::BaSchur): the factorisation then runs on the reduced camera system
void ctx_linearize(fgo_ctx *c, int buf, double *scalar_out);
void ctx_factor(fgo_ctx *c, int buf, bool with_rhs);          // with_rhs: forward solve fused into the sweep (x <- y)
void ctx_solve(fgo_ctx *c, int buf, bool fwd_done);           // backward sweep (or both), then the landmarks' back-substitution
void ba_off(fgo_ctx *c);                                      // this context gives up the eliminated form (next build: generic)
int linearize_current(fgo_ctx *c, bool want_maxdiag);

}  // namespace fgo
