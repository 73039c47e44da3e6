// Structure phase of libfgo: unique pairs, ordering, symbolic factorisation, device upload (build) and the in-place
// extension of the incremental mode (refresh_factors).
#include "fgo_ctx.hpp"
#include <memory>
#include <thread>

using namespace fgo;

namespace fgo {

namespace {
// Linearisation hubs (device_plan.hpp): the degree limit of this graph, one entry per slice of every hub variable, and the
// list of the hubs that have several slices.  deg_limit == 0: choose it.
struct HubPlan {
  int deg_limit = HUB_DEG;
  std::vector<int> var, slice;          // per entry
  std::vector<int> multi;               // 3 per multi-slice hub: variable, first entry, slices
};
void plan_hubs(const std::vector<int64_t> &he_ptr, int64_t NX, int deg_limit, HubPlan &hp) {
  static const int env_limit = (int)tune("hub_deg", 0);
  if (deg_limit <= 0 && env_limit > 0) deg_limit = env_limit;
  if (deg_limit <= 0) {
    deg_limit = HUB_DEG;
    for (int T = 64; T < HUB_DEG; T *= 2) {
      int64_t n = 0;
      for (int64_t v = 0; v < NX && n <= HUB_MAX_VARS; ++v) n += he_ptr[v + 1] - he_ptr[v] > T;
      if (n <= HUB_MAX_VARS) { deg_limit = T; break; }
    }
  }
  hp.deg_limit = deg_limit;
  hp.var.clear(); hp.slice.clear(); hp.multi.clear();
  for (int64_t v = 0; v < NX; ++v) {
    const int64_t d = he_ptr[v + 1] - he_ptr[v];
    if (d <= deg_limit) continue;
    const int ns = (int)std::min<int64_t>(HUB_MAX_SLICES, (d + HUB_SLICE - 1) / HUB_SLICE);
    if (ns > 1) { hp.multi.push_back((int)v); hp.multi.push_back((int)hp.var.size()); hp.multi.push_back(ns); }
    for (int q = 0; q < ns; ++q) { hp.var.push_back((int)v); hp.slice.push_back(q | (ns << 16)); }
  }
}
}  // namespace

static int upload_hubs(fgo_ctx *c, const HubPlan &hp, size_t entry_cap);

// IMU factors by colour.  A factor takes the smallest of 64 colours that none of its six variables carries yet, or -- a
// variable shared by more than 64 factors -- a colour of its own; the factors of one colour therefore touch disjoint H
// blocks, and a launch per colour (launch_linearize_gtsam) adds them into H without atomics, in a fixed order.  A chain of
// CombinedImuFactors (gtsam/test_ba_imu_graph.cpp:239-244: X/V/B of keyframe k shared with the next factor) takes two.
static void imu_colour_add(fgo_ctx *c, int64_t f) {
  uint64_t used = 0;
  for (int u = 0; u < 6; ++u) used |= c->imu_var_mask[(size_t)c->imu_ids[6 * f + u]];
  int col;
  if (~used) {
    col = __builtin_ctzll(~used);
    for (int u = 0; u < 6; ++u) c->imu_var_mask[(size_t)c->imu_ids[6 * f + u]] |= (uint64_t)1 << col;
  } else {
    col = 64 + c->imu_extra++;
  }
  c->imu_flist.push_back((int)f);
  c->imu_fcolor.push_back(col);
}
// colour-sorted list (stable: input order inside a colour) and its offsets
static void imu_colour_lists(fgo_ctx *c, std::vector<int> &sorted) {
  const int ncol = 64 + c->imu_extra;
  std::vector<int> cnt((size_t)ncol + 1, 0);
  for (int col : c->imu_fcolor) cnt[(size_t)col + 1]++;
  for (int k = 0; k < ncol; ++k) cnt[(size_t)k + 1] += cnt[(size_t)k];
  c->imu_color_ptr = cnt;
  sorted.resize(c->imu_flist.size());
  for (size_t i = 0; i < c->imu_flist.size(); ++i) sorted[(size_t)cnt[(size_t)c->imu_fcolor[i]]++] = c->imu_flist[i];
}

// unary priors: CSR per variable (stable in insertion order) + SoA payload with the inverse mean; `mine` selects the
// priors this rank evaluates (distributed mode)
static int upload_priors(fgo_ctx *c, int64_t NX, const std::vector<unsigned char> &mine) {
  hipStream_t s = c->stream;
  const int64_t NPall = (int64_t)c->prior_v.size();
  int64_t NP = 0;
  for (int64_t q = 0; q < NPall; ++q) NP += mine[c->prior_v[q]];
  std::vector<int64_t> prior_ptr((size_t)NX + 1, 0);
  std::vector<int> prior_pose((size_t)NP);
  std::vector<double> prior_minv((size_t)7 * NP), prior_info((size_t)21 * NP);
  for (int64_t q = 0; q < NPall; ++q) if (mine[c->prior_v[q]]) prior_ptr[c->prior_v[q] + 1]++;
  for (int64_t v = 0; v < NX; ++v) prior_ptr[v + 1] += prior_ptr[v];
  std::vector<int64_t> fill(prior_ptr.begin(), prior_ptr.end() - 1);
  for (int64_t q = 0; q < NPall; ++q) {
    if (!mine[c->prior_v[q]]) continue;
    const int64_t o = fill[c->prior_v[q]]++;
    prior_pose[o] = c->prior_v[q];
    double a[7];
    if (c->var_kind[c->prior_v[q]] == 0) pose_inv7(&c->prior_mean[(size_t)q * 7], a);
    else std::memcpy(a, &c->prior_mean[(size_t)q * 7], sizeof(a));     // vector-valued variables: raw mean
    for (int k = 0; k < 7; ++k) prior_minv[(size_t)k * NP + o] = a[k];
    for (int k = 0; k < 21; ++k) prior_info[(size_t)k * NP + o] = c->prior_info[(size_t)q * 21 + k];
  }
  HIPCHK(c, c->d_prior_ptr.upload(prior_ptr, s));
  HIPCHK(c, c->d_prior_pose.upload(prior_pose, s));
  HIPCHK(c, c->d_prior_minv.upload(prior_minv, s));
  HIPCHK(c, c->d_prior_info.upload(prior_info, s));
  HIPCHK(c, hipStreamSynchronize(s));                 // the staging vectors die here
  c->n_priors_dev = NP;
  return FGO_OK;
}

// Structure build: ordering, symbolic factorisation, device upload.  Replaces BlockSolver::buildStructure +
// the CSparse symbolic decomposition g2o redoes on iteration 0 of every optimize() call; here it is cached
// until vertices or edges are added.  The phases run in the order of the member functions below; what a phase leaves
// for the later ones are the members (until round 4 this was one 1 000-line function).
namespace {
struct StructureBuild {
  fgo_ctx *c;
  Symbolic &S;
  DevPlan &P;
  std::vector<int> &pgroup;
  explicit StructureBuild(fgo_ctx *ctx) : c(ctx), S(ctx->S), P(ctx->plan), pgroup(ctx->pose_group) {}
  struct PairRec { int a, b; int64_t e; };
  double t0 = 0;
  double t_enter = now_s();
  int64_t N = 0;
  int64_t E = 0;
  int isam_window = 0;
  int64_t R = 0;
  int64_t NX = 0;
  std::vector<int> lm_index;
  int n_lm = 0;
  std::vector<int> hidx;
  int nfree = 0;
  bool prof = 0;
  double tprev = 0;
  std::vector<PairRec> pr;
  int64_t NI = 0;
  static constexpr int64_t STRUCT_ONLY = std::numeric_limits<int64_t>::min();
  std::vector<int> ua;
  std::vector<int> ub;
  std::vector<int64_t> ufirst;
  int64_t noff = 0;
  BlockGraph g;
  std::vector<int> adj_pair;
  std::vector<int> perm;
  double t_ord0 = 0;
  double t_ord1 = 0;
  int world = 0;
  int rank = 0;
  int nb = 0;
  bool dist = 0;
  int top_col0 = 0;
  int64_t top_blk0 = 0;
  std::vector<int> pose_col;
  std::vector<int> asrc;
  std::vector<int> ptri_src;
  int byc_level = 1 << 30, byc_c0 = 0;        // row kernel, by-chunk code table: first level / first chunk that use it
  std::vector<int> prow_src;
  std::vector<int> imu_list;
  std::vector<int> edge_slot;
  std::vector<int64_t> dup_ptr;
  std::vector<int64_t> dup_edges;
  std::vector<int> dup_slot;
  std::vector<int> imu_slot;
  bool keep_lists = 0;
  std::vector<int64_t> he_ptr;
  std::vector<int> he;
  std::vector<unsigned char> var_mine;
  std::vector<int64_t> top_ext0;
  std::vector<int64_t> own_op0;
  std::vector<int64_t> own_op1;
  std::vector<int64_t> top_row0;
  std::vector<int64_t> own_row0;
  std::vector<int64_t> own_row1;
  int64_t E_cap = 0;
  int64_t NI_cap = 0;
  std::vector<double, NoInitAlloc<double>> erec;
  std::vector<int> ba_lm_var;
  std::vector<unsigned char> lm_mine;   // distributed: [n_lm] this rank eliminates the landmark (empty: all)
  std::vector<int> ba_pt_obs;
  std::vector<int> ba_obs_edge;
  std::vector<int> ba_obs_cam;
  std::vector<int> ba_obs_col;
  std::vector<int> ba_obs_lm;
  std::vector<int> ba_cam_col;
  std::vector<int> ba_tgt_blk;
  std::vector<int> ba_op_a;
  std::vector<int> ba_op_b;
  std::vector<int> ba_op_lm;
  std::vector<int64_t> ba_pt_ptr;
  std::vector<int64_t> ba_cam_ptr;
  std::vector<int64_t> ba_tgt_ptr;
  std::vector<int> ba_tgt_list;
  std::vector<int> ba_cam_list;
  std::vector<int64_t> ba_cam_t0;
  std::vector<double> ba_obs_uvw;
  int ba_n_small = 0;
  int64_t ba_o_first = 0;
  double t1 = 0;
  hipStream_t s = nullptr;
  HubPlan hubs;
  int64_t NP = 0;
  size_t hblocks = 0;
  void lap(const char *what) { if (prof) { const double t = now_s(); std::fprintf(stderr, "[fgo build]    %-28s %.1f ms\n", what, 1e3 * (t - tprev)); tprev = t; } }

  // ---- semantics of the context (g2o / GTSAM), growth reserve of the incremental mode, landmarks to eliminate, free-variable indices
  int classify() {
    t0 = now_s();
    N = (int64_t)c->ids.size(); E = (int64_t)c->ei.size();
    // one semantics per context: g2o ([t;q] tangent, VertexSE3 oplus) or GTSAM ([w;v] tangent, Expmap retraction)
    int64_t n_gtsam = 0;
    for (int64_t e = 0; e < E; ++e) n_gtsam += c->torder[e] != FGO_TANGENT_G2O;   // torder doubles as the factor kind
    bool non_pose = false;
    for (int64_t v = 0; v < N; ++v) non_pose |= c->var_kind[v] != 0;
    if (non_pose && n_gtsam != E) return fail(c, FGO_EINVAL, "plane / point / vector variables need a GTSAM-semantics graph");
    for (int64_t e = 0; e < E; ++e)
      if (c->torder[e] == 3 && !c->cam_set) return fail(c, FGO_EINVAL, "reprojection factors need fgo_set_calib_ds2 first");
    if ((n_gtsam != 0 && n_gtsam != E) || (n_gtsam == 0 && E > 0 && !c->prior_v.empty()))
      return fail(c, FGO_EINVAL, "a context holds either g2o-semantics edges or GTSAM-semantics factors, not both");
    c->gtsam_mode = n_gtsam > 0 || !c->prior_v.empty() || non_pose || !c->imu_payload.empty();
    if (!c->imu_payload.empty() && n_gtsam != E) return fail(c, FGO_EINVAL, "IMU factors need a GTSAM-semantics graph");
    if (c->dev_poses_newer) { int rc = download_poses(c); if (rc) return rc; }
    destroy_graphs(c);
    prepare_device_kernels();
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->cfg.device) == hipSuccess && cus > 0) c->sched.cus = cus; }
    // incremental mode: R phantom variables behind the real ones (free, no factors, identity diagonal)
    static const int env_reserve = std::getenv("FGO_ISAM_RESERVE") ? std::atoi(std::getenv("FGO_ISAM_RESERVE")) : 384;
    static const int env_window = std::getenv("FGO_ISAM_WINDOW") ? std::atoi(std::getenv("FGO_ISAM_WINDOW")) : 64;
    const int isam_reserve = c->isam_reserve >= 0 ? c->isam_reserve : env_reserve;
    // (g2o-semantics growth: a new key frame reaches back 1 + m_lookback_nodes <= 8 vertices, g2o/g2o_parameter.cpp:15 -- a band of
    //  16 holds it; the GTSAM drivers' IMU / plane / landmark factors get 64.  cfg 2 with band 64: +7.7 % block updates, with 16: +0.5 %)
    isam_window = c->isam_window > 0 ? c->isam_window : (c->gtsam_mode ? env_window : std::min(env_window, 16));
    // g2o-semantics graphs: the reference adds key frames between optimizeGraph() calls (g2o/test_g2o_graph.cpp:80-83); the
    // second time a structure has to be rebuilt because the graph GREW, growth mode switches itself on
    if (!c->gtsam_mode && c->grow_auto && c->built_N >= 0 && N > c->built_N && !(std::getenv("FGO_GROW") && std::atoi(std::getenv("FGO_GROW")) == 0))
      c->grow_incremental = true;
    const bool growing = c->gtsam_mode ? c->isam_incremental : c->grow_incremental;
    R = (growing && c->shard_world == 1) ? isam_reserve : 0;
    NX = N + R;
    c->inc.valid = false;
    c->n_phantom = (int)R;
    // ---- bundle adjustment: free Point3 variables that carry reprojection factors (and unary priors) only are eliminated
    // analytically (kernels_ba.hip, device_plan.hpp "BaPlan") instead of becoming columns of the block system
    const int64_t NI_all = (int64_t)c->imu_payload.size();
    lm_index = std::vector<int>((size_t)NX, -1);
    n_lm = 0;
    {
      const int ba_on = std::getenv("FGO_BA_SCHUR") ? std::atoi(std::getenv("FGO_BA_SCHUR")) : 1;       // (read per build: tests switch it)
      const int ba_min = std::getenv("FGO_BA_MIN") ? std::atoi(std::getenv("FGO_BA_MIN")) : 1000;
      // (distributed mode: every rank eliminates the SAME set -- the structure is replicated -- and takes the landmarks of its own
      //  domain's cameras, lm_mine below)
      if (ba_on && !c->ba_disable && c->gtsam_mode && !c->isam_incremental && c->cam_set) {
        std::vector<char> ok((size_t)N, 0);
        std::vector<int> deg((size_t)N, 0);
        for (int64_t v = 0; v < N; ++v) ok[v] = c->var_kind[v] == 2 && !c->fixed[v];
        for (int64_t e = 0; e < E; ++e) {
          if (c->torder[e] == 3) { ok[c->ei[e]] = 0; deg[c->ej[e]]++; }
          else { ok[c->ei[e]] = 0; ok[c->ej[e]] = 0; }
        }
        for (int64_t f = 0; f < NI_all; ++f) for (int u = 0; u < 6; ++u) ok[c->imu_ids[6 * f + u]] = 0;
        int cnt = 0;
        for (int64_t v = 0; v < N; ++v) cnt += ok[v] && deg[v] > 0;
        if (cnt >= ba_min) {
          // numbered by the first camera that sees them (then by id): a camera's observations are stored by landmark number, so
          // the landmarks of neighbouring lanes of the per-landmark kernels then sit next to each other in every per-observation
          // array (k_ba_back read 2.1 GB for 0.72 GB of W with the landmarks in id order of a generator that scatters them; a front
          // end that creates landmarks keyframe by keyframe -- gtsam/gtsam_graph.cpp:387-394 -- has this order anyway)
          std::vector<int> first_cam((size_t)N, INT32_MAX);
          for (int64_t e = 0; e < E; ++e) if (c->torder[e] == 3 && ok[c->ej[e]]) first_cam[c->ej[e]] = std::min(first_cam[c->ej[e]], c->ei[e]);
          std::vector<int64_t> start((size_t)N + 1, 0);
          for (int64_t v = 0; v < N; ++v) if (ok[v] && deg[v] > 0) start[(size_t)first_cam[v] + 1]++;
          for (int64_t v = 0; v < N; ++v) start[(size_t)v + 1] += start[(size_t)v];
          for (int64_t v = 0; v < N; ++v) if (ok[v] && deg[v] > 0) { lm_index[v] = (int)start[(size_t)first_cam[v]]++; ++n_lm; }
        }
      }
  }
  // free-variable (hessian) index per pose
  hidx = std::vector<int>((size_t)NX, -1);
  nfree = 0;
  for (int64_t v = 0; v < NX; ++v) if ((v >= N || !c->fixed[v]) && lm_index[v] < 0) hidx[v] = nfree++;
  if (nfree == 0 && n_lm > 0) {                    // nothing but landmarks is free (pure triangulation): they stay columns
    std::fill(lm_index.begin(), lm_index.end(), -1);
    n_lm = 0;
    for (int64_t v = 0; v < NX; ++v) if (v >= N || !c->fixed[v]) hidx[v] = nfree++;
  }
  if (nfree == 0 || (E == 0 && c->prior_v.empty() && c->imu_payload.empty()))
    return fail(c, FGO_ESTATE, "nothing to optimise (no free vertex or no factor)");
  prof = std::getenv("FGO_SYM_PROFILE") != nullptr;
  tprev = now_s();
    return FGO_OK;
  }

  // ---- unique variable pairs of all factors (+ co-visibility pairs of eliminated landmarks, + the phantom band) and the block graph
  int pairs_and_graph() {
    // unique vertex pairs
    {   // binary factors: count per chunk of edges, prefix sums, fill (the edge order is kept)
      constexpr int64_t CH = 1 << 16;
      const int nch = (int)((E + CH - 1) / CH);
      std::vector<int64_t> at((size_t)nch + 1, 0);
      auto valid = [&](int64_t e, int &a, int &b) { a = hidx[c->ei[e]]; b = hidx[c->ej[e]]; return a >= 0 && b >= 0 && a != b; };
      parallel_ranges(nch, 1, [&](int c0, int c1) {
        for (int ch = c0; ch < c1; ++ch) {
          int64_t n = 0; int a, b;
          for (int64_t e = ch * CH, e1 = std::min(E, e + CH); e < e1; ++e) n += valid(e, a, b);
          at[(size_t)ch + 1] = n;
        }
      });
      for (int ch = 0; ch < nch; ++ch) at[(size_t)ch + 1] += at[(size_t)ch];
      pr.resize((size_t)at[(size_t)nch]);
      parallel_ranges(nch, 1, [&](int c0, int c1) {
        for (int ch = c0; ch < c1; ++ch) {
          int64_t w = at[(size_t)ch]; int a, b;
          for (int64_t e = ch * CH, e1 = std::min(E, e + CH); e < e1; ++e) if (valid(e, a, b)) pr[(size_t)w++] = {std::min(a, b), std::max(a, b), e};
        }
      });
  }
  // the 6-variable IMU factors contribute all 15 variable pairs; encoded as e = -1 - (15 f + pair)
  NI = (int64_t)c->imu_payload.size();
  for (int64_t f = 0; f < NI; ++f) {
    int q = 0;
    for (int u = 0; u < 6; ++u)
      for (int w = u + 1; w < 6; ++w, ++q) {
        const int a = hidx[c->imu_ids[6 * f + u]], b = hidx[c->imu_ids[6 * f + w]];
        if (a < 0 || b < 0 || a == b) continue;
        pr.push_back({std::min(a, b), std::max(a, b), -1 - (15 * f + q)});
      }
  }
  // (STRUCT_ONLY: class constant)   // a pair without a factor (yet)
  if (n_lm > 0) {
    // eliminating a landmark couples all the cameras that see it: the co-visibility pairs, found per camera (in parallel)
    // from its landmarks' camera lists and de-duplicated there
    std::vector<int64_t> lc_ptr((size_t)n_lm + 1, 0), cl_ptr((size_t)nfree + 1, 0);
    for (int64_t e = 0; e < E; ++e)
      if (c->torder[e] == 3 && lm_index[c->ej[e]] >= 0 && hidx[c->ei[e]] >= 0) { lc_ptr[lm_index[c->ej[e]] + 1]++; cl_ptr[hidx[c->ei[e]] + 1]++; }
    for (int p = 0; p < n_lm; ++p) lc_ptr[p + 1] += lc_ptr[p];
    for (int a = 0; a < nfree; ++a) cl_ptr[a + 1] += cl_ptr[a];
    std::vector<int> lc((size_t)lc_ptr[n_lm]), cl((size_t)cl_ptr[nfree]);       // cameras of a landmark / landmarks of a camera
    {
      std::vector<int64_t> f1(lc_ptr.begin(), lc_ptr.end() - 1), f2(cl_ptr.begin(), cl_ptr.end() - 1);
      for (int64_t e = 0; e < E; ++e)
        if (c->torder[e] == 3 && lm_index[c->ej[e]] >= 0 && hidx[c->ei[e]] >= 0) {
          const int p = lm_index[c->ej[e]], a = hidx[c->ei[e]];
          lc[f1[p]++] = a; cl[f2[a]++] = p;
        }
    }
    std::vector<std::vector<int>> nbrs((size_t)nfree);
    parallel_ranges(nfree, 64, [&](int a0, int a1) {
      std::vector<int> tmp;
      for (int a = a0; a < a1; ++a) {
        tmp.clear();
        for (int64_t q = cl_ptr[a]; q < cl_ptr[a + 1]; ++q) {
          const int p = cl[q];
          for (int64_t w = lc_ptr[p]; w < lc_ptr[p + 1]; ++w) if (lc[w] > a) tmp.push_back(lc[w]);
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        nbrs[a] = tmp;
      }
    });
    for (int a = 0; a < nfree; ++a) for (int b : nbrs[a]) pr.push_back({a, b, STRUCT_ONLY});
  }
  for (int64_t k = 0; k < R; ++k)                                            // phantom k couples to the `window` variables before it
    for (int64_t u = std::max<int64_t>(0, N + k - isam_window); u < N + k; ++u)
      if (hidx[u] >= 0) pr.push_back({std::min(hidx[u], hidx[N + k]), std::max(hidx[u], hidx[N + k]), STRUCT_ONLY});
  {   // sort by (a, b, e): counting sort on a (counts and places claimed with atomic increments: the order inside a run does
      // not matter), then the (short) runs of equal a are sorted in parallel
    std::vector<int64_t> start((size_t)nfree + 1, 0);
    const int npr = (int)std::min<size_t>(pr.size(), (size_t)INT32_MAX);
    parallel_ranges(npr, 1 << 16, [&](int i0, int i1) { for (int i = i0; i < i1; ++i) __atomic_fetch_add(&start[(size_t)pr[(size_t)i].a + 1], (int64_t)1, __ATOMIC_RELAXED); });
    for (size_t i = (size_t)npr; i < pr.size(); ++i) start[(size_t)pr[i].a + 1]++;
    for (int i = 0; i < nfree; ++i) start[i + 1] += start[i];
    std::vector<PairRec> sorted(pr.size());
    {
      std::vector<int64_t> fill(start.begin(), start.end() - 1);
      parallel_ranges(npr, 1 << 16, [&](int i0, int i1) {
        for (int i = i0; i < i1; ++i) sorted[(size_t)__atomic_fetch_add(&fill[(size_t)pr[(size_t)i].a], (int64_t)1, __ATOMIC_RELAXED)] = pr[(size_t)i];
      });
      for (size_t i = (size_t)npr; i < pr.size(); ++i) sorted[(size_t)fill[(size_t)pr[i].a]++] = pr[i];
    }
    parallel_ranges(nfree, 4096, [&](int ab, int ae) {
      for (int a = ab; a < ae; ++a)
        std::sort(sorted.begin() + start[a], sorted.begin() + start[a + 1],
                  [](const PairRec &x, const PairRec &y) { return x.b != y.b ? x.b < y.b : x.e < y.e; });
    });
    pr.swap(sorted);
  }
  // unique pairs
  // index in pr of the first member
  {   // heads of the runs of equal (a, b): count per chunk, prefix sums, fill
    constexpr int64_t CH = 1 << 16;
    const int64_t np = (int64_t)pr.size();
    const int nch = (int)((np + CH - 1) / CH);
    auto head = [&](int64_t i) { return i == 0 || pr[(size_t)i].a != pr[(size_t)i - 1].a || pr[(size_t)i].b != pr[(size_t)i - 1].b; };
    std::vector<int64_t> at((size_t)nch + 1, 0);
    parallel_ranges(nch, 1, [&](int c0, int c1) {
      for (int ch = c0; ch < c1; ++ch) {
        int64_t n = 0;
        for (int64_t i = ch * CH, i1 = std::min(np, i + CH); i < i1; ++i) n += head(i);
        at[(size_t)ch + 1] = n;
      }
    });
    for (int ch = 0; ch < nch; ++ch) at[(size_t)ch + 1] += at[(size_t)ch];
    const int64_t nu = at[(size_t)nch];
    ua.resize((size_t)nu); ub.resize((size_t)nu); ufirst.resize((size_t)nu + 1);
    parallel_ranges(nch, 1, [&](int c0, int c1) {
      for (int ch = c0; ch < c1; ++ch) {
        int64_t w = at[(size_t)ch];
        for (int64_t i = ch * CH, i1 = std::min(np, i + CH); i < i1; ++i)
          if (head(i)) { ua[(size_t)w] = pr[(size_t)i].a; ub[(size_t)w] = pr[(size_t)i].b; ufirst[(size_t)w] = i; ++w; }
      }
    });
    ufirst[(size_t)nu] = np;
  }
  noff = (int64_t)ua.size();
  c->n_offdiag = noff;
  g.n = nfree;
  g.xadj.assign((size_t)nfree + 1, 0);
  // pair index of every adjacency entry, in the order the block graph lists them
  {   // adjacency lists: a vertex's neighbours ascending (what the serial fill over the sorted pairs produced), places claimed
      // with atomic increments and every list sorted afterwards -- as (neighbour, pair) keys, so that adj_pair comes with it
    const int nh = (int)std::min<int64_t>(noff, INT32_MAX);
    parallel_ranges(nh, 1 << 16, [&](int h0, int h1) {
      for (int h = h0; h < h1; ++h) { __atomic_fetch_add(&g.xadj[(size_t)ua[(size_t)h] + 1], 1, __ATOMIC_RELAXED); __atomic_fetch_add(&g.xadj[(size_t)ub[(size_t)h] + 1], 1, __ATOMIC_RELAXED); }
    });
    if (noff > nh) return fail(c, FGO_EINVAL, "more than 2^31 block pairs");
    for (int i = 0; i < nfree; ++i) g.xadj[i + 1] += g.xadj[i];
    std::vector<uint64_t> key((size_t)g.xadj[nfree]);
    std::vector<int> fill(g.xadj.begin(), g.xadj.end() - 1);
    parallel_ranges(nh, 1 << 16, [&](int h0, int h1) {
      for (int h = h0; h < h1; ++h) {
        key[(size_t)__atomic_fetch_add(&fill[(size_t)ua[(size_t)h]], 1, __ATOMIC_RELAXED)] = ((uint64_t)(uint32_t)ub[(size_t)h] << 32) | (uint32_t)h;
        key[(size_t)__atomic_fetch_add(&fill[(size_t)ub[(size_t)h]], 1, __ATOMIC_RELAXED)] = ((uint64_t)(uint32_t)ua[(size_t)h] << 32) | (uint32_t)h;
      }
    });
    g.adj.resize(key.size()); adj_pair.resize(key.size());
    parallel_ranges(nfree, 4096, [&](int v0, int v1) {
      for (int v = v0; v < v1; ++v) {
        std::sort(key.begin() + g.xadj[v], key.begin() + g.xadj[v + 1]);
        for (int p = g.xadj[v]; p < g.xadj[v + 1]; ++p) { g.adj[(size_t)p] = (int)(key[(size_t)p] >> 32); adj_pair[(size_t)p] = (int)(uint32_t)key[(size_t)p]; }
      }
    });
  }
  lap("pairs + block graph");
    return FGO_OK;
  }

  // ---- nested dissection, symbolic factorisation, schedule (ordering.cpp, symbolic.cpp)
  int order_and_symbolic() {
    const char *wl = std::getenv("FGO_TASK_WORK");
    // light subtrees (one workgroup each, level 0): flat optimum 1250 .. 10000 on cfg 2 since the panel kernels exist
    const int64_t work_limit = wl ? std::atoll(wl) : 5000;
    const double cl_v = tune("chain_work", -1.0);
    // chains become panels (<= PANEL_MAX columns); with the LDS panel kernels the work bound no longer pays
    // (cfg 2: 60000 -> 31.6 it/s, unbounded -> 38.7 it/s)
    const int64_t chain_limit = cl_v >= 0 ? (int64_t)cl_v : (int64_t)1 << 60;
    world = c->shard_world; rank = c->shard_rank;
    // The bisection is greedy and the landscape of (levels, fill) noisy: on 100k-pose graphs the variants below differ by
    // 1-4 levels and 2-8 % iterations/s, and which one wins depends on the graph.  fgo_config.order_candidates (or
    // FGO_TUNE=nd_try=N) > 1 builds the first N of them and keeps the structure with the lowest PREDICTED sweep time -- a model
    // fitted on the measured sweeps (profiles/NOTES.md round 4): 76 us per level (every level is a chain of dependent launches
    // whatever it holds) + 0.051 ns per block update.  One candidate (the default) costs nothing extra.
    // Distributed mode: the cut into sub-trees per rank wants the better-balanced bisections of weight 8 (per-rank trial of
    // cfg 2 at 8 ranks, tools/dist_rank_timing.py: 2.21 ms against 2.55 with weight 5; one GPU: 148 against 151 it/s).
    struct Cand { double bal_w; int leaf; };
    static const Cand cands1[4] = {{5.0, 64}, {4.0, 64}, {8.0, 64}, {4.0, 96}}, candsN[4] = {{8.0, 64}, {5.0, 64}, {12.0, 64}, {8.0, 96}};
    const Cand *cands = world > 1 ? candsN : cands1;
    const int n_try = std::max(1, std::min(4, c->cfg.order_candidates > 0 ? c->cfg.order_candidates : (int)tune("nd_try", 1)));
    t_ord0 = now_s();
    double t_ord = 0, best_cost = 0;
    int best = -1;
    for (int q = 0; q < n_try; ++q) {
      OrderingOptions oo;
      oo.leaf = c->cfg.nd_leaf > 0 ? c->cfg.nd_leaf : (int)tune("nd_leaf", cands[q].leaf);
      oo.bal_w = tune("nd_bal_w", cands[q].bal_w); oo.bal_t = tune("nd_bal_t", oo.bal_t);
      oo.dense_factor = tune("dense_factor", oo.dense_factor);
      // (time dissection ignores the balance weight: its candidates vary the balance window instead -- five 100k-pose seeds, predicted
      //  sweep + backward of the best of four against the first: -2 % on average, profiles/NOTES.md round 5)
      static const double tcand[4][2] = {{0.30, 0.0}, {0.30, 0.1}, {0.35, 0.0}, {0.25, 0.1}};
      oo.time_side = tune("nd_time_side", tcand[q][0]); oo.time_weight = tune("nd_time_weight", tcand[q][1]);
      // (recovered time labels: graphs whose variables are all poses -- a pose graph read from a file with arbitrary ids; ordering.cpp)
      oo.time_recover = true;
      for (int64_t v = 0; v < N && oo.time_recover; ++v) oo.time_recover = c->var_kind[(size_t)v] == 0;
      std::vector<int> pq;
      const double ta = now_s();
      nested_dissection(g, oo, pq);
      t_ord += now_s() - ta;
      if ((int)pq.size() != nfree) return fail(c, FGO_EINVAL, "internal: ordering lost vertices");
      if (n_try == 1) { perm.swap(pq); S.cus = c->sched.cus; build_symbolic(g, perm, work_limit, chain_limit, S, world); best = 0; break; }
      Symbolic Sq;
      Sq.cus = c->sched.cus;
      build_symbolic(g, pq, work_limit, chain_limit, Sq, world, true);      // (analysis only: the winner is built in full below)
      const double cost = 76.0 * (double)(Sq.level_ptr.size() - 1) + 0.051e-3 * (double)Sq.nops;      // us
      if (prof) std::fprintf(stderr, "[fgo build]    ordering candidate %d (balance %.1f, leaf %d): %zu levels, %lld block updates, nnz(L) %lld -> predicted %.0f us\n", q,
                             oo.bal_w, oo.leaf, Sq.level_ptr.size() - 1, (long long)Sq.nops, (long long)Sq.nnzL, cost);
      if (best < 0 || cost < best_cost) { best = q; best_cost = cost; perm.swap(pq); }
    }
    if (n_try > 1) { S.cus = c->sched.cus; build_symbolic(g, perm, work_limit, chain_limit, S, world); }
    t_ord1 = t_ord0 + t_ord;
    lap("ordering + build_symbolic");
    nb = nfree;
    dist = world > 1;
    top_col0 = dist ? S.dom_col0[world] : nb;
    top_blk0 = dist ? S.colptr[top_col0] : S.nnzL;
    return FGO_OK;
  }

  // ---- L -> H map, panel sources, factor ownership (distributed), H slots, half-edge lists, edge records, landmark-elimination tables
  int host_tables() {
    // pose -> elimination position
    pose_col = std::vector<int>((size_t)NX, -1);
    for (int64_t v = 0; v < NX; ++v) if (hidx[v] >= 0) pose_col[v] = S.iperm[hidx[v]];
    for (int64_t v = 0; v < N; ++v) if (lm_index[v] >= 0) pose_col[v] = nb + lm_index[v];      // eliminated landmarks: virtual columns (b / x only)
    // L block -> H block: column k's original entries are the graph neighbours of perm[k]; stamp them in a scratch row
    // (per host thread) and read the column's pattern against it
    asrc = std::vector<int>((size_t)S.nnzL, -1);
    {
      // one chunk per host thread: the scratch rows are allocated once per chunk
      parallel_ranges(nb, std::max(2048, (nb + host_threads() - 1) / host_threads()), [&](int kb, int ke) {
        std::vector<int> stamp((size_t)nb, -1), pair_of((size_t)nb, -1);
        for (int k = kb; k < ke; ++k) {
          const int ha = S.perm[k];
          for (int p = g.xadj[ha]; p < g.xadj[ha + 1]; ++p) { const int col = S.iperm[g.adj[p]]; stamp[col] = k; pair_of[col] = adj_pair[p]; }
          asrc[S.colptr[k]] = k;
          for (int64_t t = S.colptr[k] + 1; t < S.colptr[k + 1]; ++t) {
            const int i = S.rowidx[t];
            asrc[t] = stamp[i] == k ? nb + pair_of[i] : -1;
          }
        }
      });
  }
  lap("asrc");
  // panel blocks: where a block's value sits when the panel kernels pick it up -- in L (>= 0: block id; the wide
  // accumulate kernel already applied its external updates), still in H (-2 - H block), or nowhere (-1: fill-in
  // without updates).  Structural, so resolved here instead of by three dependent loads per block on the device.
  auto block_src = [&](int t) -> int {
    if (t < 0) return -1;
    if (t >= top_blk0) return t;                  // distributed: a top block's value arrives in L through the collective
    if (S.op_mid[t] > S.op_ptr[t]) return t;
    return asrc[t] >= 0 ? -2 - asrc[t] : -1;
  };
  ptri_src = std::vector<int>(S.ptri_blk.size()); prow_src = std::vector<int>(S.prow_blk.size());
  parallel_ranges((int)std::min<size_t>(S.ptri_blk.size(), INT32_MAX), 1 << 16, [&](int q0, int q1) { for (int q = q0; q < q1; ++q) ptri_src[q] = block_src(S.ptri_blk[q]); });
  parallel_ranges((int)std::min<size_t>(S.prow_blk.size(), INT32_MAX), 1 << 16, [&](int q0, int q1) { for (int q = q0; q < q1; ++q) prow_src[q] = block_src(S.prow_blk[q]); });
  lap("panel sources");
  // multi-GPU: which factors this context linearises (everything when world == 1).  A variable belongs to the rank whose
  // domain holds its column (group `world` = top, -1 = fixed); a factor to the rank of any of its domain variables (they
  // all lie in one domain: a factor is a clique of the block graph and domains are separated by the top), factors among
  // top / fixed variables only are dealt round-robin.  So a domain variable sees ALL its factors locally (complete
  // diagonal block), a top variable a partial sum -- completed by the collective on the tail of L.
  pgroup.assign((size_t)NX, -1);
  if (dist)
    for (int64_t v = 0; v < N; ++v)
      if (pose_col[v] >= 0) pgroup[v] = (int)(std::upper_bound(S.dom_col0.begin(), S.dom_col0.begin() + world + 1, pose_col[v]) - S.dom_col0.begin()) - 1;
  auto factor_owner = [&](const int *vars, int nv, int64_t salt) -> int {
    if (!dist) return 0;
    int own = -1;
    for (int q = 0; q < nv; ++q) { const int gq = pgroup[vars[q]]; if (gq >= 0 && gq < world) { if (own >= 0 && own != gq) return -2; own = gq; } }
    return own >= 0 ? own : (int)(salt % world);
  };
  // eliminated landmarks: the cameras of one landmark are a clique of the reduced graph (co-visibility pairs), i.e. they lie in ONE
  // domain plus the top -- the rank of that domain eliminates the landmark (all of its observations, its prior, its step); landmarks
  // seen from top cameras only are dealt round-robin.  Its variable counts as that rank's (unary terms, LM scale, final gather).
  lm_mine.clear();
  if (dist && n_lm > 0) {
    std::vector<int> owner((size_t)n_lm, -1);
    for (int64_t e = 0; e < E; ++e) {
      if (c->torder[e] != 3) continue;
      const int p = lm_index[c->ej[e]];
      if (p < 0) continue;
      const int gq = pgroup[c->ei[e]];
      if (gq >= 0 && gq < world) {
        if (owner[p] >= 0 && owner[p] != gq) return fail(c, FGO_EINVAL, "internal: the cameras of an eliminated landmark span two domains");
        owner[p] = gq;
      }
    }
    lm_mine.assign((size_t)n_lm, 0);
    for (int64_t v = 0; v < N; ++v) {
      const int p = lm_index[v];
      if (p < 0) continue;
      if (owner[p] < 0) owner[p] = p % world;
      pgroup[v] = owner[p];
      lm_mine[p] = owner[p] == rank;
    }
  }
  std::vector<unsigned char> edge_mine((size_t)E, 1), imu_mine((size_t)NI, 1);
  if (dist) {
    for (int64_t e = 0; e < E; ++e) {
      const int vars[2] = {c->ei[e], c->ej[e]};
      const int o = factor_owner(vars, 2, e);
      if (o == -2) return fail(c, FGO_EINVAL, "internal: a factor spans two domains");
      edge_mine[e] = o == rank;
    }
    for (int64_t f = 0; f < NI; ++f) {
      const int o = factor_owner(&c->imu_ids[6 * f], 6, f);
      if (o == -2) return fail(c, FGO_EINVAL, "internal: an IMU factor spans two domains");
      imu_mine[f] = o == rank;
    }
  }
  // this rank's IMU factors, sorted by colour
  c->imu_var_mask.assign((size_t)NX, 0); c->imu_flist.clear(); c->imu_fcolor.clear(); c->imu_extra = 0;
  for (int64_t f = 0; f < NI; ++f) if (imu_mine[f]) imu_colour_add(c, f);
  imu_colour_lists(c, imu_list);
  // edge -> slot; duplicate groups
  edge_slot = std::vector<int>((size_t)E, -1);
  dup_ptr = std::vector<int64_t>{0};
  imu_slot = std::vector<int>((size_t)15 * NI, -1);
  {
    // per pair: the slot of its single binary factor / of its IMU entries (in parallel); pairs that carry several binary
    // factors (rare) are collected per chunk and get their duplicate groups in pair order afterwards
    constexpr int64_t CH = 1 << 15;
    const int nch = (int)((noff + CH - 1) / CH);
    std::vector<std::vector<int64_t>> multi((size_t)nch);
    parallel_ranges(nch, 1, [&](int c0, int c1) {
      for (int ch = c0; ch < c1; ++ch)
        for (int64_t h = ch * CH, h1 = std::min(noff, h + CH); h < h1; ++h) {
          const int64_t m0 = ufirst[h], m1 = ufirst[h + 1];
          int64_t nbin = 0;
          for (int64_t m = m0; m < m1; ++m) nbin += pr[m].e >= 0;
          if (nbin > 1) multi[(size_t)ch].push_back(h);
          for (int64_t m = m0; m < m1; ++m) {
            const int64_t e = pr[m].e;
            if (e == STRUCT_ONLY) continue;
            if (e < 0) {                                  // IMU pair (u < w): stored transposed when w is eliminated later
              const int64_t idx = -1 - e, f = idx / 15;
              int u = 0, w = 1;
              for (int q = (int)(idx % 15); q > 0; --q) { if (++w == 6) { ++u; w = u + 1; } }
              const int cu = pose_col[c->imu_ids[6 * f + u]], cw = pose_col[c->imu_ids[6 * f + w]];
              imu_slot[idx] = (int)(((nb + h) << 1) | (cw > cu ? 1 : 0));
              continue;
            }
            if (nbin == 1) {
              const int ci = pose_col[c->ei[e]], cj = pose_col[c->ej[e]];
              edge_slot[e] = (int)(((nb + h) << 1) | (cj > ci ? 1 : 0));
            }
          }
        }
    });
    for (int ch = 0; ch < nch; ++ch)
      for (const int64_t h : multi[(size_t)ch]) {
        for (int64_t m = ufirst[h]; m < ufirst[h + 1]; ++m) {
          const int64_t e = pr[m].e;
          if (e < 0) continue;                            // (STRUCT_ONLY is negative too)
          const int ci = pose_col[c->ei[e]], cj = pose_col[c->ej[e]];
          if (edge_mine[e]) { dup_edges.push_back(e); dup_slot.push_back((int)(((nb + h) << 1) | (cj > ci ? 1 : 0))); }   // owned members only
        }
        if ((int64_t)dup_edges.size() > dup_ptr.back()) dup_ptr.push_back((int64_t)dup_edges.size());
      }
  }
  lap("edge slots");
  if (R > 0) {      // what refresh_factors needs to append factors / claim phantom slots without touching the structure
    fgo_ctx::Incr &I = c->inc;
    I.NX = NX; I.N_done = N; I.E_done = E; I.NI_done = NI; I.NP_done = (int64_t)c->prior_v.size(); I.nb = nb;
    I.hidx = hidx; I.pose_col = pose_col;
    I.ukey.resize((size_t)noff);
    for (int64_t h = 0; h < noff; ++h) I.ukey[h] = ((uint64_t)(uint32_t)ua[h] << 32) | (uint32_t)ub[h];
    I.edge_h.assign((size_t)E, -1);
    I.pair_nbin.assign((size_t)noff, 0); I.pair_first.assign((size_t)noff, -1);
    I.dups.clear();
    for (int64_t h = 0; h < noff; ++h)
      for (int64_t m = ufirst[h]; m < ufirst[h + 1]; ++m) {
        const int64_t e = pr[m].e;
        if (e < 0) continue;                         // IMU pair or structure-only
        I.edge_h[e] = (int)h;
        if (I.pair_nbin[h]++ == 0) I.pair_first[h] = (int)e;
      }
    for (int64_t h = 0; h < noff; ++h)
      if (I.pair_nbin[h] > 1)
        for (int64_t m = ufirst[h]; m < ufirst[h + 1]; ++m) if (pr[m].e >= 0) I.dups[(int)h].push_back(pr[m].e);
    I.edge_slot = edge_slot;
  }
  keep_lists = R > 0;
  // per-variable incidence of the IMU factors
  std::vector<int64_t> imu_inc_ptr((size_t)NX + 1, 0);
  std::vector<int> imu_inc((size_t)6 * imu_list.size());
  {
    for (int f : imu_list) for (int u = 0; u < 6; ++u) imu_inc_ptr[c->imu_ids[6 * (int64_t)f + u] + 1]++;
    for (int64_t v = 0; v < NX; ++v) imu_inc_ptr[v + 1] += imu_inc_ptr[v];
    std::vector<int64_t> fill(imu_inc_ptr.begin(), imu_inc_ptr.end() - 1);
    for (int f : imu_list)
      for (int u = 0; u < 6; ++u) imu_inc[fill[c->imu_ids[6 * (int64_t)f + u]]++] = (int)(((int64_t)f << 3) | u);
  }
  // half-edge lists (owned edges only)
  he_ptr = std::vector<int64_t>((size_t)NX + 1, 0);
  int64_t n_mine = 0;
  // (both sides of an eliminated observation are linearised by kernels_ba.hip: k_ba_linearize / k_ba_cameras)
  // (counts and places by atomic increments on all host threads; a variable's list is then sorted: ascending edge, which is the
  //  order the serial fill produced and the order its contributions are summed in)
  const int En = (int)std::min<int64_t>(E, INT32_MAX);
  parallel_ranges(En, 1 << 16, [&](int e0, int e1) {
    int64_t mine = 0;
    for (int64_t e = e0; e < e1; ++e)
      if (edge_mine[e]) {
        if (lm_index[c->ej[e]] < 0) { __atomic_fetch_add(&he_ptr[(size_t)c->ei[e] + 1], (int64_t)1, __ATOMIC_RELAXED); __atomic_fetch_add(&he_ptr[(size_t)c->ej[e] + 1], (int64_t)1, __ATOMIC_RELAXED); }
        ++mine;
      }
    __atomic_fetch_add(&n_mine, mine, __ATOMIC_RELAXED);
  });
  for (int64_t v = 0; v < NX; ++v) he_ptr[v + 1] += he_ptr[v];
  he = std::vector<int>((size_t)2 * n_mine);
  {
    std::vector<int64_t> fill(he_ptr.begin(), he_ptr.end() - 1);
    parallel_ranges(En, 1 << 16, [&](int e0, int e1) {
      for (int64_t e = e0; e < e1; ++e) {
        if (!edge_mine[e]) continue;
        if (lm_index[c->ej[e]] >= 0) continue;
        he[(size_t)__atomic_fetch_add(&fill[(size_t)c->ei[e]], (int64_t)1, __ATOMIC_RELAXED)] = (int)(e << 1);
        he[(size_t)__atomic_fetch_add(&fill[(size_t)c->ej[e]], (int64_t)1, __ATOMIC_RELAXED)] = (int)((e << 1) | 1);
      }
    });
    parallel_ranges((int)std::min<int64_t>(NX, INT32_MAX), 4096, [&](int v0, int v1) {
      for (int v = v0; v < v1; ++v) if (he_ptr[v + 1] - he_ptr[v] > 1) std::sort(he.begin() + he_ptr[v], he.begin() + he_ptr[v + 1]);
    });
  }
  // unary terms (priors, the padding identity of 3-dof variables): the variable's rank; top / fixed variables: rank 0
  var_mine = std::vector<unsigned char>((size_t)NX, 1);
  if (dist) for (int64_t v = 0; v < N; ++v) var_mine[v] = (pgroup[v] >= 0 && pgroup[v] < world) ? pgroup[v] == rank : rank == 0;
  // distributed: per top block / top column, where the updates sourced from this rank's domain and from the top start
  if (dist) {
    const int64_t ntb = S.nnzL - top_blk0;
    const int ntc = nb - top_col0;
    const int lo = S.dom_col0[rank], hi = S.dom_col0[rank + 1];
    top_ext0.resize((size_t)ntb); own_op0.resize((size_t)ntb); own_op1.resize((size_t)ntb);
    top_row0.resize((size_t)ntc); own_row0.resize((size_t)ntc); own_row1.resize((size_t)ntc);
    parallel_ranges((int)std::min<int64_t>(ntb, INT32_MAX), 4096, [&](int qb, int qe) {
      for (int64_t q = qb; q < qe; ++q) {
        const int64_t t = top_blk0 + q;
        const int *a0 = S.op_a.data() + S.op_ptr[t], *a1 = S.op_a.data() + S.op_mid[t];     // external ops, ascending source column
        auto first_col_ge = [&](int col) { return (int64_t)(std::partition_point(a0, a1, [&](int blk) { return S.blkcol[blk] < col; }) - S.op_a.data()); };
        own_op0[q] = first_col_ge(lo); own_op1[q] = first_col_ge(hi); top_ext0[q] = first_col_ge(top_col0);
      }
    });
    for (int q = 0; q < ntc; ++q) {
      const int k = top_col0 + q;
      const int *r0 = S.row_col.data() + S.rowptr[k], *r1 = S.row_col.data() + S.row_mid[k];   // entries outside the column's own panel, ascending
      auto first_ge = [&](int col) { return (int64_t)(std::lower_bound(r0, r1, col) - S.row_col.data()); };
      own_row0[q] = first_ge(lo); own_row1[q] = first_ge(hi); top_row0[q] = first_ge(top_col0);
    }
  }
  lap("half-edge lists");
  if (keep_lists) { c->inc.he_ptr = he_ptr; c->inc.he = he; c->inc.imu_inc_ptr = imu_inc_ptr; c->inc.imu_inc = imu_inc; }
  // edge payload: one 256-byte record per edge (device_plan.hpp EDGE_REC)
  // (incremental mode: room for factors that arrive later)
  E_cap = R > 0 ? E + std::max<int64_t>(4096, E / 8) : E;
  NI_cap = R > 0 ? NI + std::max<int64_t>(256, NI / 8) : NI;
  erec = std::vector<double, NoInitAlloc<double>>((size_t)EDGE_REC * E_cap);   // first touched by the threads that fill it
  parallel_ranges((int)std::min<int64_t>(E, INT32_MAX), 8192, [&](int eb, int ee) {
    for (int64_t e = eb; e < ee; ++e) {
      double *o = &erec[(size_t)EDGE_REC * e];
      if (c->torder[e] <= 1) pose_inv7(&c->meas[(size_t)e * 7], o);          // SE3 factors: inverse measurement
      else std::memcpy(o, &c->meas[(size_t)e * 7], 7 * sizeof(double));     // plane / reprojection: raw payload
      o[7] = 0.0;
      std::memcpy(o + 8, &c->info[(size_t)e * 21], 21 * sizeof(double));
      o[29] = o[30] = o[31] = 0.0;
    }
  });
  lap("edge records");
  // ---- landmark elimination tables (device_plan.hpp "BaPlan")
  ba_n_small = 0;
  ba_o_first = 0;
  if (n_lm > 0) {
    ba_lm_var.resize((size_t)n_lm);
    for (int64_t v = 0; v < N; ++v) if (lm_index[v] >= 0) ba_lm_var[lm_index[v]] = (int)v;
    // observations camera-major: counting sort on the camera's column (fixed cameras, column -1, first), landmarks ascending inside
    std::vector<int64_t> cstart((size_t)nb + 2, 0);
    int64_t n_obs = 0;
    auto obs_here = [&](int64_t e) { return c->torder[e] == 3 && lm_index[c->ej[e]] >= 0 && (lm_mine.empty() || lm_mine[lm_index[c->ej[e]]]); };
    for (int64_t e = 0; e < E; ++e) if (obs_here(e)) { cstart[pose_col[c->ei[e]] + 2]++; ++n_obs; }
    for (int k = 0; k <= nb; ++k) cstart[k + 1] += cstart[k];
    ba_obs_edge.resize((size_t)n_obs);
    {
      std::vector<int64_t> fill(cstart.begin(), cstart.end() - 1);
      for (int64_t e = 0; e < E; ++e) if (obs_here(e)) ba_obs_edge[fill[pose_col[c->ei[e]] + 1]++] = (int)e;
    }
    parallel_ranges(nb + 1, 16, [&](int k0, int k1) {
      for (int k = k0; k < k1; ++k)
        std::sort(ba_obs_edge.begin() + cstart[k], ba_obs_edge.begin() + cstart[k + 1], [&](int x, int y) {
          const int px = lm_index[c->ej[x]], py = lm_index[c->ej[y]];
          return px != py ? px < py : x < y; });
    });
    lap("  ba: observations camera-major");
    ba_obs_cam.resize((size_t)n_obs); ba_obs_col.resize((size_t)n_obs); ba_obs_lm.resize((size_t)n_obs);
    ba_pt_ptr.assign((size_t)n_lm + 1, 0);
    parallel_ranges((int)std::min<int64_t>(n_obs, INT32_MAX), 1 << 16, [&](int ob, int oe) {
      for (int64_t o = ob; o < oe; ++o) {
        const int e = ba_obs_edge[o];
        ba_obs_cam[o] = c->ei[e]; ba_obs_col[o] = pose_col[c->ei[e]]; ba_obs_lm[o] = lm_index[c->ej[e]];
      }
    });
    for (int64_t o = 0; o < n_obs; ++o) ba_pt_ptr[ba_obs_lm[o] + 1]++;
    for (int p = 0; p < n_lm; ++p) ba_pt_ptr[p + 1] += ba_pt_ptr[p];
    ba_pt_obs.resize((size_t)n_obs);
    {
      std::vector<int64_t> fill(ba_pt_ptr.begin(), ba_pt_ptr.end() - 1);
      for (int64_t o = 0; o < n_obs; ++o) ba_pt_obs[fill[ba_obs_lm[o]]++] = (int)o;       // ascending o = ascending column
    }
    lap("  ba: per-observation arrays, landmark lists");
    ba_cam_ptr.push_back(cstart[1]);
    for (int k = 0; k < nb; ++k) if (cstart[k + 2] > cstart[k + 1]) { ba_cam_col.push_back(k); ba_cam_ptr.push_back(cstart[k + 2]); }
    // blocks of the reduced system with landmark terms, per column camera k: for its observation o and every observation o2 of
    // the same landmark from a camera of column >= k: block (col(o2), k) -= Y(o2) Y(o)^T
    std::vector<int64_t> ustart((size_t)nfree + 1, 0);
    for (int64_t h = 0; h < noff; ++h) ustart[ua[h] + 1]++;
    for (int a = 0; a < nfree; ++a) ustart[a + 1] += ustart[a];
    const int ncam = (int)ba_cam_col.size();
    // two passes over the (observation, observation of the same landmark from a later-or-equal column) pairs of every column
    // camera, both in the same order: count per (camera, row column), prefix sums, fill -- no sorting, no growing vectors.
    // Within a block the pairs keep the emission order (the camera's observations ascending, then the landmark's list).
    std::vector<std::vector<int>> cam_rows((size_t)ncam);          // distinct row columns of a camera, ascending
    std::vector<std::vector<int64_t>> cam_cnt((size_t)ncam);       // pairs per row column
    std::atomic<int> missing{0};
    auto for_each_pair = [&](int i, auto &&fn) {
      const int k = ba_cam_col[i];
      for (int64_t o = ba_cam_ptr[i]; o < ba_cam_ptr[i + 1]; ++o) {
        const int p = ba_obs_lm[o];
        for (int64_t q = ba_pt_ptr[p]; q < ba_pt_ptr[p + 1]; ++q) {
          const int o2 = ba_pt_obs[q];
          if (ba_obs_col[o2] >= k) fn(ba_obs_col[o2], o2, (int)o);
        }
      }
    };
    parallel_ranges(ncam, 4, [&](int i0, int i1) {
      static thread_local std::vector<int> slot;                   // column -> local row index + 1 (0: unseen); reset after use
      if ((int)slot.size() < nb) slot.assign((size_t)nb, 0);
      for (int i = i0; i < i1; ++i) {
        std::vector<int> &rows = cam_rows[(size_t)i];
        for_each_pair(i, [&](int row, int, int) { if (!slot[row]) { slot[row] = 1; rows.push_back(row); } });
        std::sort(rows.begin(), rows.end());
        for (size_t x = 0; x < rows.size(); ++x) slot[rows[x]] = (int)x + 1;
        std::vector<int64_t> &cnt = cam_cnt[(size_t)i];
        cnt.assign(rows.size(), 0);
        for_each_pair(i, [&](int row, int, int) { cnt[slot[row] - 1]++; });
        for (int r : rows) slot[r] = 0;
      }
    });
    lap("  ba: pair counts");
    std::vector<int64_t> t0v((size_t)ncam + 1, 0), e0v((size_t)ncam + 1, 0);
    for (int i = 0; i < ncam; ++i) {
      t0v[i + 1] = t0v[i] + (int64_t)cam_rows[i].size();
      int64_t n = 0;
      for (int64_t x : cam_cnt[i]) n += x;
      e0v[i + 1] = e0v[i] + n;
    }
    ba_tgt_blk.resize((size_t)t0v[ncam]); ba_tgt_ptr.assign((size_t)t0v[ncam] + 1, 0);
    ba_op_a.resize((size_t)e0v[ncam]); ba_op_b.resize((size_t)e0v[ncam]); ba_op_lm.resize((size_t)e0v[ncam]);
    parallel_ranges(ncam, 4, [&](int i0, int i1) {
      static thread_local std::vector<int> slot;
      if ((int)slot.size() < nb) slot.assign((size_t)nb, 0);
      std::vector<int64_t> cur;
      for (int i = i0; i < i1; ++i) {
        const int k = ba_cam_col[i];
        const std::vector<int> &rows = cam_rows[(size_t)i];
        cur.resize(rows.size());
        int64_t at = e0v[i];
        for (size_t x = 0; x < rows.size(); ++x) {
          const int row = rows[x];
          slot[row] = (int)x + 1;
          cur[x] = at;
          at += cam_cnt[(size_t)i][x];
          ba_tgt_ptr[(size_t)t0v[i] + x + 1] = at;
          int blk = -1;
          if (row == k) blk = k;
          else {
            const int ha = std::min(S.perm[k], S.perm[row]), hb = std::max(S.perm[k], S.perm[row]);
            const int *b0 = ub.data() + ustart[ha], *b1 = ub.data() + ustart[ha + 1];
            const int *f = std::lower_bound(b0, b1, hb);
            if (f != b1 && *f == hb) blk = nb + (int)(f - ub.data());
          }
          if (blk < 0) missing++;
          ba_tgt_blk[(size_t)t0v[i] + x] = blk;
        }
        for_each_pair(i, [&](int row, int o2, int o) { const int64_t w = cur[slot[row] - 1]++; ba_op_a[w] = o2; ba_op_b[w] = o; ba_op_lm[w] = ba_obs_lm[o]; });
        for (int r : rows) slot[r] = 0;
      }
    });
    if (missing.load() > 0) return fail(c, FGO_EINVAL, "internal: a co-visibility pair has no block in the reduced system");
    lap("  ba: pair lists");
    // k_ba_schur_cam takes the blocks of a COLUMN camera together (its observations' B = (H_pp + lambda I)^-1 W^T staged in LDS
    // once for all of the camera's blocks); a camera with more row blocks than the kernel has accumulator slots keeps the
    // block-by-block kernel: short lists first (one wave per block), long ones behind (four waves)
    {
      static const int small_max = (int)tune("ba_small", 80);
      static const int cam_on = (int)tune("ba_schur_cam", 1);
      constexpr int CAM_SLOTS = 80;                                 // kernels_ba.hip: lane groups of a k_ba_schur_cam workgroup
      ba_cam_t0 = t0v;
      for (int i = 0; i < ncam; ++i) {
        const bool staged = cam_on && t0v[i + 1] - t0v[i] <= CAM_SLOTS;
        if (staged) ba_cam_list.push_back(i);
        else for (int64_t t = t0v[i]; t < t0v[i + 1]; ++t) ba_tgt_list.push_back((int)t);
      }
      auto mid = std::stable_partition(ba_tgt_list.begin(), ba_tgt_list.end(), [&](int t) { return ba_tgt_ptr[t + 1] - ba_tgt_ptr[t] <= small_max; });
      ba_n_small = (int)(mid - ba_tgt_list.begin());
    }
    ba_obs_uvw.resize(3 * (size_t)n_obs);
    parallel_ranges((int)std::min<int64_t>(n_obs, INT32_MAX), 1 << 16, [&](int ob, int oe) {
      for (int64_t o = ob; o < oe; ++o) {
        const int e = ba_obs_edge[o];
        ba_obs_uvw[3 * o] = c->meas[(size_t)e * 7]; ba_obs_uvw[3 * o + 1] = c->meas[(size_t)e * 7 + 1]; ba_obs_uvw[3 * o + 2] = c->info[(size_t)e * 21];
      }
    });
    ba_o_first = cstart[1];
    if (prof) std::fprintf(stderr, "[fgo build]    landmark elimination: %d landmarks, %lld observations, %d cameras, %zu reduced blocks (%d with <= 80 pairs), %zu pairs\n",
                           n_lm, (long long)n_obs, ncam, ba_tgt_blk.size(), ba_n_small, ba_op_a.size());
    lap("landmark elimination tables");
  }
  t1 = now_s();

    return FGO_OK;
  }

  // ---- device buffers: index lists, edge records, hubs, priors, IMU payloads, panel tables, chain / BA tables, work buffers
  int upload() {
    // ---- upload
    s = c->stream;
    HIPCHK(c, c->d_pose_col.upload(pose_col, s));
    if (R > 0) {     // capacity first (upload() keeps an allocation that is large enough), so that later factors are appended in place
      HIPCHK(c, c->d_edge_i.alloc((size_t)E_cap)); HIPCHK(c, c->d_edge_j.alloc((size_t)E_cap)); HIPCHK(c, c->d_edge_slot.alloc((size_t)E_cap));
      HIPCHK(c, c->d_edge_kind.alloc((size_t)E_cap)); HIPCHK(c, c->d_he.alloc((size_t)2 * E_cap));
      HIPCHK(c, c->d_imu.alloc((size_t)NI_cap)); HIPCHK(c, c->d_imu_ids.alloc((size_t)6 * NI_cap)); HIPCHK(c, c->d_imu_slot.alloc((size_t)15 * NI_cap));
      HIPCHK(c, c->d_imu_list.alloc((size_t)NI_cap));
  }
  HIPCHK(c, c->d_edge_i.upload(c->ei, s));
  HIPCHK(c, c->d_edge_j.upload(c->ej, s));
  HIPCHK(c, c->d_edge_slot.upload(edge_slot, s));
  HIPCHK(c, c->d_he_ptr.upload(he_ptr, s));
  plan_hubs(he_ptr, NX, 0, hubs);
  const size_t hub_cap = hubs.var.size() + (R > 0 ? 256 : 0);       // entries the scratch buffers have room for
  { const int rc = upload_hubs(c, hubs, hub_cap); if (rc) return rc; }
  if (keep_lists) { c->inc.hub_deg = hubs.deg_limit; c->inc.hub_cap = hub_cap; }
  HIPCHK(c, c->d_he.upload(he, s));
  HIPCHK(c, c->d_dup_ptr.upload(dup_ptr, s));
  HIPCHK(c, c->d_dup_edges.upload(dup_edges, s));
  HIPCHK(c, c->d_dup_slot.upload(dup_slot, s));
  HIPCHK(c, c->d_ainv.upload(erec, s));
  { const int rc = upload_priors(c, NX, var_mine); if (rc) return rc; }
  NP = c->n_priors_dev;
  HIPCHK(c, c->d_imu.upload(c->imu_payload, s));
  HIPCHK(c, c->d_imu_ids.upload(c->imu_ids, s));
  HIPCHK(c, c->d_imu_slot.upload(imu_slot, s));
  {
    std::vector<int> vk(c->var_kind);
    vk.resize((size_t)NX, 5);                          // phantoms: kind 5 = no degrees of freedom yet (identity block, x = 0)
    HIPCHK(c, c->d_var_kind.upload(vk, s));
    HIPCHK(c, hipStreamSynchronize(s));
  }
  HIPCHK(c, c->d_edge_kind.upload(c->torder, s));
  HIPCHK(c, c->d_imu_list.upload(imu_list, s));
  HIPCHK(c, c->d_pose_group.upload(pgroup, s));
  HIPCHK(c, c->d_var_mine.upload(var_mine, s));
  HIPCHK(c, c->d_top_ext0.upload(top_ext0, s));
  HIPCHK(c, c->d_own_op0.upload(own_op0, s));
  HIPCHK(c, c->d_own_op1.upload(own_op1, s));
  HIPCHK(c, c->d_top_row0.upload(top_row0, s));
  HIPCHK(c, c->d_own_row0.upload(own_row0, s));
  HIPCHK(c, c->d_own_row1.upload(own_row1, s));
  if (dist) HIPCHK(c, c->d_gather.alloc(std::max<size_t>((size_t)N * 8, 64)));
  HIPCHK(c, c->d_colptr.upload(S.colptr, s));
  HIPCHK(c, c->d_rowidx.upload(S.rowidx, s));
  HIPCHK(c, c->d_asrc.upload(asrc, s));
  HIPCHK(c, c->d_op_ptr.upload(S.op_ptr, s));
  HIPCHK(c, c->d_op_mid.upload(S.op_mid, s));
  HIPCHK(c, c->d_op_a.upload(S.op_a, s));
  HIPCHK(c, c->d_op_b.upload(S.op_b, s));
  HIPCHK(c, c->d_acc_targets.upload(S.acc_targets, s));
  for (auto &a : S.g2_a) if (a < 0) a = (int)S.nnzL;        // absent (row, source) pairs read the zero block
  HIPCHK(c, c->d_g2_tgt.upload(S.g2_tgt, s));
  HIPCHK(c, c->d_g2_ptr.upload(S.g2_ptr, s));
  HIPCHK(c, c->d_g2_b.upload(S.g2_b, s));
  HIPCHK(c, c->d_g2_a.upload(S.g2_a, s));
  HIPCHK(c, c->d_ride_items.upload(S.ride_items, s));
  HIPCHK(c, c->d_acc_start.upload(S.acc_start, s));
  {
    // one descriptor per accumulate target: the head of k_chol_acc read acc_targets / acc_start, THEN op_ptr / op_mid / asrc / top_ext0 of the
    // block -- two dependent round trips in a launch that is five of them on the narrow levels
    std::vector<AccDesc> ad(S.acc_targets.size());
    parallel_ranges((int)ad.size(), 1 << 14, [&](int q0, int q1) {
      for (int q = q0; q < q1; ++q) {
        const int t = S.acc_targets[(size_t)q];
        const int64_t rsv = S.acc_start.empty() ? -1 : S.acc_start[(size_t)q];
        const bool from_h = rsv >= 0 && ((rsv >> 62) & 1);
        const int64_t rs = rsv >= 0 ? (rsv & ~((int64_t)1 << 62)) : -1;
        const bool topb = dist && t >= top_blk0;
        ad[(size_t)q] = AccDesc{t, (topb || (rs >= 0 && !from_h)) ? -2 : (asrc[(size_t)t] >= 0 ? asrc[(size_t)t] : -1),
                                (long long)(rs >= 0 ? rs : (topb ? top_ext0[(size_t)(t - top_blk0)] : S.op_ptr[(size_t)t])), (long long)S.op_mid[(size_t)t]};
      }
    });
    HIPCHK(c, c->d_acc_desc.upload(ad, s));
  }
  c->isam_L_valid = false;
  c->col_task.clear();
  if (c->isam_incremental && !dist) {               // partial sweeps: task of every column / accumulate target / column group
    const int ntask = (int)S.task_ptr.size() - 1;
    c->col_task.assign((size_t)nb, 0);
    std::vector<int> tcol_task((size_t)nb);
    for (int t = 0; t < ntask; ++t)
      for (int q = S.task_ptr[t]; q < S.task_ptr[t + 1]; ++q) { c->col_task[S.task_cols[q]] = t; tcol_task[(size_t)q] = t; }
    std::vector<int> acc_task(S.acc_targets.size()), g2_task(S.g2_ptr.size() - 1);
    for (size_t q = 0; q < acc_task.size(); ++q) acc_task[q] = c->col_task[S.blkcol[S.acc_targets[q]]];
    for (size_t q = 0; q < g2_task.size(); ++q) {
      int t = -1;
      for (int x = 0; x < ACC2_G && t < 0; ++x) if (S.g2_tgt[q * ACC2_G + x] >= 0) t = c->col_task[S.blkcol[S.g2_tgt[q * ACC2_G + x]]];
      g2_task[q] = t < 0 ? 0 : t;
    }
    // per task: where its items sit in the level-wise lists (partial sweeps launch the covering ranges only)
    {
      const int nl = (int)S.level_ptr.size() - 1;
      c->tk_s0.assign((size_t)ntask, 0); c->tk_s1 = c->tk_s0; c->tk_l0 = c->tk_s0; c->tk_l1 = c->tk_s0;
      c->tk_g0.assign((size_t)ntask, 0); c->tk_g1 = c->tk_g0; c->tk_c0 = c->tk_g0; c->tk_c1 = c->tk_g0;
      c->task_level.assign((size_t)ntask, 0);
      c->lvl_lo.assign((size_t)nl, 0); c->lvl_hi.assign((size_t)nl, -1);
      bool ok = true;
      // items [b, e) of a level, task(q) of an item (-1: belongs to whatever task is current) -> per task [first, end)
      const char *kind = "";
      auto scan = [&](int l, int64_t b, int64_t e, auto task_at, auto &first, auto &end) {
        const bool was_ok = ok;
        int64_t q = b;
        for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) {
          first[(size_t)t] = (typename std::decay<decltype(first)>::type::value_type)q;
          while (q < e) { const int tq = task_at(q); if (tq != t && tq >= 0) break; ++q; }
          end[(size_t)t] = (typename std::decay<decltype(end)>::type::value_type)q;
        }
        if (q != e) ok = false;
        if (was_ok && !ok && prof) std::fprintf(stderr, "[fgo build]    %s of level %d not in task order: item %lld of [%lld, %lld) has task %d, level tasks [%d, %d)\n", kind, l, (long long)q, (long long)b, (long long)e, task_at(q), S.level_ptr[l], S.level_ptr[l + 1]);
      };
      for (int l = 0; l < nl; ++l) {
        for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) c->task_level[(size_t)t] = l;
        kind = "short targets";
        scan(l, S.acc_ptr[l], S.acc_mid[l], [&](int64_t q) { return acc_task[(size_t)q]; }, c->tk_s0, c->tk_s1);
        kind = "long targets";
        scan(l, S.acc_mid[l], S.acc_ptr[l + 1], [&](int64_t q) { return acc_task[(size_t)q]; }, c->tk_l0, c->tk_l1);
        kind = "column groups";
        if (!S.g2_lvl.empty())
          scan(l, S.g2_lvl[l], S.g2_lvl[l + 1], [&](int64_t q) {
            for (int x = 0; x < ACC2_G; ++x) if (S.g2_tgt[(size_t)q * ACC2_G + x] >= 0) return c->col_task[S.blkcol[S.g2_tgt[(size_t)q * ACC2_G + x]]];
            return -1; }, c->tk_g0, c->tk_g1);
        kind = "row chunks";
        scan(l, S.rchunk_ptr[l], S.rchunk_ptr[l + 1], [&](int64_t q) { return S.panel_task[S.rchunk_panel[(size_t)q]]; }, c->tk_c0, c->tk_c1);
      }
      c->tk_ok = ok && tune("isam_ranges", 1) != 0;
    }
    HIPCHK(c, c->d_acc_task.upload(acc_task, s));
    HIPCHK(c, c->d_g2_task.upload(g2_task, s));
    HIPCHK(c, c->d_tcol_task.upload(tcol_task, s));
    HIPCHK(c, c->d_task_dirty.alloc((size_t)ntask));
    HIPCHK(c, c->d_col_dirty.alloc((size_t)nb));
    HIPCHK(c, c->d_moved.alloc((size_t)NX));
    HIPCHK(c, c->d_moved_next.alloc((size_t)NX));
    HIPCHK(c, c->d_lin_mask.alloc((size_t)NX));
    HIPCHK(c, c->d_chi_var.alloc((size_t)NX));
    HIPCHK(c, c->d_xprev.alloc((size_t)nb * 6));
    HIPCHK(c, c->d_bwd_run.alloc((size_t)ntask));
    HIPCHK(c, c->d_chg.alloc((size_t)nb));
    c->wild_valid = false;
    c->isam_moved_valid = false;                    // (the flags of the previous update were laid out for the previous structure)
    c->isam_H_valid = false;
    HIPCHK(c, c->d_y.alloc((size_t)nb * 6));
    {
      const size_t need = 3 * (size_t)NX + (size_t)ntask + (size_t)nb;
      if (need > c->h_flags_cap) {
        if (c->h_flags) (void)hipHostFree(c->h_flags);
        c->h_flags = nullptr; c->h_flags_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->h_flags, need + need / 4, hipHostMallocDefault));
        c->h_flags_cap = need + need / 4;
      }
      std::memset(c->h_flags, 0, need);             // fgo_isam2_update keeps the flag arrays clean between calls (sparse set / clear)
      c->isam_set_tasks.clear(); c->isam_set_cols.clear(); c->isam_set_aff.clear();
    }
    HIPCHK(c, hipStreamSynchronize(s));             // the staging vectors die here
  }
  HIPCHK(c, c->d_rowptr.upload(S.rowptr, s));
  HIPCHK(c, c->d_row_blk.upload(S.row_blk, s));
  HIPCHK(c, c->d_row_col.upload(S.row_col, s));
  HIPCHK(c, c->d_task_ptr.upload(S.task_ptr, s));
  HIPCHK(c, c->d_task_cols.upload(S.task_cols, s));
  HIPCHK(c, c->d_task_panel.upload(S.task_panel, s));
  HIPCHK(c, c->d_panel_task.upload(S.panel_task, s));
  HIPCHK(c, c->d_ptri_blk.upload(S.ptri_blk, s));
  HIPCHK(c, c->d_prow_ptr.upload(S.prow_ptr, s));
  HIPCHK(c, c->d_prow_idx.upload(S.prow_idx, s));
  HIPCHK(c, c->d_prow_blk.upload(S.prow_blk, s));
  HIPCHK(c, c->d_pchunk_panel.upload(S.pchunk_panel, s));
  HIPCHK(c, c->d_pchunk_row0.upload(S.pchunk_row0, s));
  HIPCHK(c, c->d_pchunk_nrows.upload(S.pchunk_nrows, s));
  HIPCHK(c, c->d_panel_chunk0.upload(S.panel_chunk0, s));
  HIPCHK(c, c->d_row_mid.upload(S.row_mid, s));
  HIPCHK(c, c->d_fchunk_col.upload(S.fchunk_col, s));
  HIPCHK(c, c->d_fchunk_e0.upload(S.fchunk_e0, s));
  HIPCHK(c, c->d_pcol_fchunk0.upload(S.pcol_fchunk0, s));
  HIPCHK(c, c->d_pcol_fchunkn.upload(S.pcol_fchunkn, s));
  {
    std::vector<PanelDesc> pd((size_t)S.n_panels);
    std::vector<int> task_level((size_t)S.task_ptr.size() - 1, 0);
    for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l)
      for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) task_level[t] = (int)l;
    // operand tiles only for panels that run the panel kernels; PanelDesc::top = the panel's first tile (of 256 doubles) in ptop:
    // NJ (NJ + 1) / 2 = 21 tiles for the NJ = ceil(6 x 16 / 16) tile rows of a 16-column panel
    int64_t n_tiles = 0;
    for (int pn = 0; pn < S.n_panels; ++pn) {
      const int t = S.panel_task[pn];
      const int rows = S.prow_ptr[pn + 1] - S.prow_ptr[pn];
      constexpr int NJ = (6 * PANEL_MAX + 15) / 16;
      const bool has = S.level_panel[task_level[t]];
      pd[pn] = PanelDesc{t, S.task_ptr[t + 1] - S.task_ptr[t], S.task_ptr[t], S.prow_ptr[pn], rows, S.panel_chunk0[pn],
                         (rows + PANEL_ROWS - 1) / PANEL_ROWS, has ? (int)n_tiles : -1};
      if (has) n_tiles += NJ * (NJ + 1) / 2;
    }
    if (n_tiles > INT32_MAX) return fail(c, FGO_ENOMEM, "operand-tile table too large");
    HIPCHK(c, c->d_ptop.alloc((size_t)n_tiles * 256));
    std::vector<RowChunk> rc(S.rchunk_panel.size());
    for (size_t q = 0; q < rc.size(); ++q) {
      const PanelDesc &d = pd[S.rchunk_panel[q]];
      rc[q] = RowChunk{S.rchunk_panel[q], d.m, S.rchunk_s0[q], 6 * d.nrows, d.prow0, d.cols0, d.top, d.task};
    }
    std::vector<BwdChunk> bc(S.pchunk_panel.size());
    for (size_t q = 0; q < bc.size(); ++q) bc[q] = BwdChunk{S.pchunk_panel[q], pd[S.pchunk_panel[q]].m, S.pchunk_row0[q], S.pchunk_nrows[q]};
    HIPCHK(c, c->d_pdesc.upload(pd, s));
    {
      // Launch order of the throughput triangle kernels (k_panel_tri1, k_panel_tri<8>) within a level: by panel WIDTH instead of
      // task order.  A wide level is 1.5-3 rounds of panel workgroups, a panel's time goes with its column count, and k_panel_tri1
      // is 79 KB of straight-line code for a 64 KB instruction cache that two CUs share -- neighbours of equal width run in step
      // and share its lines.  Measured on cfg 2 (us per launch, levels 1 / 2 / 4 / 5; task order 123 / 96 / 62 / 44): widest first
      // 113 / 105 / 47 / 35, narrowest first 105 / 99 / 51 / 37, alternating 148 / 101 / 72 / 45.  -> one wave per panel
      // (k_panel_tri1): narrowest first; eight waves (k_panel_tri<8>): widest first.  FGO_TUNE tri_lpt = 0 task order, 2 / 3 force.
      std::vector<int> order((size_t)S.n_panels);
      for (int pn = 0; pn < S.n_panels; ++pn) order[(size_t)pn] = pn;
      const int mode = (int)tune("tri_lpt", 1);
      if (mode != 0)
        for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l) {
          const int ntf = S.level_ptr[l + 1] - S.level_ptr[l];
          if (!S.level_panel[l] || ntf <= 0) continue;
          const int p0 = S.task_panel[S.level_ptr[l]], p1 = p0 + ntf;
          if (p0 < 0 || p1 > S.n_panels) continue;
          const bool tri1 = tune("tri1", 1) != 0 && ntf > tri_wide_panels(c->sched.cus) && ntf >= (int)tune("tri1_min", 3 * c->sched.cus);   // (launch_factor's choice)
          const bool ascending = mode == 2 || (mode == 1 && tri1);
          if (ascending) std::stable_sort(order.begin() + p0, order.begin() + p1, [&](int a, int b) { return pd[(size_t)a].m < pd[(size_t)b].m; });
          else std::stable_sort(order.begin() + p0, order.begin() + p1, [&](int a, int b) { return pd[(size_t)a].m > pd[(size_t)b].m; });
        }
      HIPCHK(c, c->d_tri_order.upload(order, s));
    }
    {
      // k_chol_leaf: one descriptor per light sub-tree instead of the chain task -> task_ptr -> task_cols -> colptr -> op_ptr (four
      // dependent round trips at the head of a workgroup that lives ~40 us), and the tasks of a leaf level by DESCENDING work for full
      // sweeps (the last round of a launch -- 3.6 rounds of 768 workgroups at cfg 2 -- is then the light ones)
      const int ntask = (int)S.task_ptr.size() - 1;
      std::vector<LeafDesc> ld((size_t)ntask, LeafDesc{0, 0, 0, 0, 0, 0, 0, 0}), lpt;
      for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l) {
        if (l >= S.level_leaf.size() || !S.level_leaf[l]) continue;
        for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) {
          const int cb = S.task_ptr[t], m = S.task_ptr[t + 1] - cb, k0 = S.task_cols[cb];
          const int64_t base = S.colptr[k0], nblk = S.colptr[k0 + m] - base, ob = S.op_ptr[base];
          ld[(size_t)t] = LeafDesc{t, cb, m, k0, (long long)base, (long long)ob, (int)nblk, (int)(S.op_ptr[base + nblk] - ob)};
        }
      }
      lpt = ld;
      if (tune("leaf_lpt", 1) != 0)
        for (size_t l = 0; l + 1 < S.level_ptr.size(); ++l)
          if (l < S.level_leaf.size() && S.level_leaf[l])
            std::stable_sort(lpt.begin() + S.level_ptr[l], lpt.begin() + S.level_ptr[l + 1], [](const LeafDesc &a, const LeafDesc &b) { return a.nops + 2 * a.nblk > b.nops + 2 * b.nblk; });
      HIPCHK(c, c->d_leaf_desc.upload(ld, s));
      HIPCHK(c, c->d_leaf_lpt.upload(lpt, s));
    }
    HIPCHK(c, c->d_rchunks.upload(rc, s));
    {
      // NARROW levels (a single round of few row waves: a pure chain of dependent round trips): the source codes of a chunk's 16 scalar rows in a
      // table laid out BY CHUNK, addressed by the chunk index alone and so requested beside the chunk's descriptor instead of after it.  (On the wide
      // levels the 16 code rows per chunk cost more cache lines than the round trip is worth: measured, profiles/NOTES.md.)
      const int thr = (int)tune("rows_byc", 600);                      // row chunks per level below which the table is used (0: never)
      const int nl = (int)S.rchunk_ptr.size() - 1;
      int l0 = nl;
      while (l0 > 0 && S.rchunk_ptr[l0] - S.rchunk_ptr[l0 - 1] < thr) --l0;
      byc_level = l0 < nl && thr > 0 ? l0 : (1 << 30);
      byc_c0 = byc_level < nl ? S.rchunk_ptr[byc_level] : (int)rc.size();
      std::vector<int> rsrc((rc.size() - (size_t)byc_c0) * 16 * PANEL_MAX, -1);
      for (size_t q = (size_t)byc_c0; q < rc.size(); ++q) {
        const RowChunk &r = rc[q];
        for (int nn = 0; nn < 16; ++nn) {
          const int sr = r.s0 + nn;
          int *dst = rsrc.data() + ((q - (size_t)byc_c0) * 16 + nn) * PANEL_MAX;
          if (sr < r.R6) { const size_t ro = S.row_off(r.prow0 + sr / 6); for (int k = 0; k < r.m; ++k) dst[k] = prow_src[ro + k]; }
          else if (sr == r.R6) for (int k = 0; k < r.m; ++k) dst[k] = S.task_cols[r.cols0 + k];
        }
      }
      if (!rsrc.empty()) HIPCHK(c, c->d_rchunk_src.upload(rsrc, s)); else c->d_rchunk_src.release();
    }
    // backward chain (k_bwd_chain): the top levels of the tree -- from the root level down while a level consists of panels and
    // has few of them -- run in ONE launch, a workgroup per panel in top-down order, each waiting for the panels above
    {
      static const int chain_on = (int)tune("bwd_chain", 1);
      static const int chain_max = (int)tune("bwd_chain_max", 64);   // panels per level
      std::vector<ChainItem> items;
      const int nl = (int)S.level_ptr.size() - 1;
      int low = nl, chain_levels = 0;
      // (distributed: the segments of the replicated top only -- a rank's own segments run after it, level by level)
      static const int chain_dist = (int)tune("bwd_chain_dist", 1);
      if (chain_on && (world == 1 || chain_dist) && !std::getenv("FGO_NO_PANELS"))
        for (int l = nl - 1; l >= 1; --l) {
          if (world > 1 && S.seg_group[l] != world) continue;
          const int nt = S.level_ptr[l + 1] - S.level_ptr[l];
          if (!S.level_panel[l] || nt > chain_max || nt == 0) break;
          const int need = (int)items.size();
          for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) items.push_back(ChainItem{S.task_panel[t], need});
          low = l;
          ++chain_levels;
        }
      // a single level gains nothing.  (Counted in LEVELS placed in the chain, not in schedule segments between `low` and the root:
      //  distributed, the domains' segments interleave with the top's and said nothing about the chain's length -- ADVICE r5)
      if (chain_levels < 2) { items.clear(); low = -1; }
      c->sched.bchain_low = items.empty() ? -1 : low;
      c->sched.bchain_n = (int)items.size();
      HIPCHK(c, c->d_bchain.upload(items, s));
      HIPCHK(c, c->d_bchain_done.alloc(1));
      HIPCHK(c, hipMemsetAsync(c->d_bchain_done.p, 0, sizeof(unsigned), s));
    }
    HIPCHK(c, c->d_bchunks.upload(bc, s));
    HIPCHK(c, hipStreamSynchronize(s));       // the staging vectors die at the end of this scope
  }
  HIPCHK(c, c->d_ptri_src.upload(ptri_src, s));
  HIPCHK(c, c->d_prow_src.upload(prow_src, s));
  HIPCHK(c, c->d_rchunk_panel.upload(S.rchunk_panel, s));
  HIPCHK(c, c->d_rchunk_s0.upload(S.rchunk_s0, s));
  HIPCHK(c, c->d_fpart.alloc(S.fchunk_col.size() * 6));
  HIPCHK(c, c->d_bpart.alloc(S.pchunk_panel.size() * PANEL_MAX * 6));
  hblocks = (size_t)nb + (size_t)noff;
  for (int i = 0; i < 2; ++i) {
    HIPCHK(c, c->d_poses[i].alloc((size_t)NX * 8));
    if (R > 0) HIPCHK(c, hipMemsetAsync(c->d_poses[i].p + (size_t)N * 8, 0, sizeof(double) * (size_t)R * 8, s));
    HIPCHK(c, c->d_H[i].alloc(hblocks * 36));
    HIPCHK(c, c->d_b[i].alloc(((size_t)nb + (size_t)n_lm) * 6));          // (+ the virtual columns of eliminated landmarks)
    HIPCHK(c, hipMemsetAsync(c->d_H[i].p, 0, sizeof(double) * hblocks * 36, s));
  }
  HIPCHK(c, c->d_x.alloc(((size_t)nb + (size_t)n_lm) * 6));
  // behind the factor: the zero block (nnzL), an identity block (nnzL + 1) and the scratch blocks of hub targets' rider pieces
  HIPCHK(c, c->d_L.alloc(((size_t)S.nnzL + 2 + (size_t)S.n_scratch) * 36));
  HIPCHK(c, hipMemsetAsync(c->d_L.p + (size_t)S.nnzL * 36, 0, sizeof(double) * 36 * (2 + (size_t)S.n_scratch), s));
  {
    static const double ident[36] = {1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1};
    HIPCHK(c, hipMemcpyAsync(c->d_L.p + ((size_t)S.nnzL + 1) * 36, ident, sizeof(ident), hipMemcpyHostToDevice, s));
  }
  HIPCHK(c, c->d_scal.alloc(8));
  HIPCHK(c, c->d_fail.alloc(1));
  HIPCHK(c, hipMemsetAsync(c->d_fail.p, 0, sizeof(int), s));
  // two-pass reduction scratch, sized from the real launch shapes: linearise = ceil(4N/256) lane-group workgroups + one
  // per hub variable (bounded by 2E / HUB_DEG, NOT by N / HUB_DEG) + one per IMU factor; chi2 <= 2048 + ceil(NI/64);
  // maxdiag <= 1024; update / relinearise ceil(N/256)
  const size_t npart = std::max<size_t>({(size_t)4096, (size_t)((NX * 4 + 255) / 256) + hub_cap + (size_t)NI_cap + 64 + (size_t)((n_lm + 255) / 256),
                                         (size_t)2048 + (size_t)((NI_cap + 63) / 64) + 64, (size_t)((NX + 255) / 256) + 64});
  HIPCHK(c, c->d_imu_stash.alloc((size_t)NI_cap * 150));
  HIPCHK(c, c->d_partial.alloc(npart));
  HIPCHK(c, hipStreamSynchronize(s));

  // landmark elimination: tables and buffers
  {
    fgo_ctx::BaSchur &ba = c->ba;
    ba.on = n_lm > 0; ba.n_lm = n_lm;
    BaPlan &B = c->plan.ba;
    std::memset(&B, 0, sizeof(B));
    if (n_lm > 0) {
      const size_t n_obs = ba_obs_edge.size();
      HIPCHK(c, ba.d_lm_var.upload(ba_lm_var, s)); HIPCHK(c, ba.d_pt_ptr.upload(ba_pt_ptr, s)); HIPCHK(c, ba.d_pt_obs.upload(ba_pt_obs, s));
      HIPCHK(c, ba.d_obs_uvw.upload(ba_obs_uvw, s)); HIPCHK(c, ba.d_tgt_list.upload(ba_tgt_list, s));
      HIPCHK(c, ba.d_cam_t0.upload(ba_cam_t0, s)); HIPCHK(c, ba.d_cam_list.upload(ba_cam_list, s));
      // the same measurements in the landmarks' order (k_ba_linearize: one lane per landmark streams its observations instead of
      // gathering 24 bytes from a different cache line each -- the PMC pass showed 1.6 GB of reads for 0.15 GB of data)
      std::vector<double> pt_uvw(3 * n_obs);
      std::vector<int> pt_cam(n_obs);
      parallel_ranges((int)std::min<size_t>(n_obs, (size_t)INT32_MAX), 1 << 16, [&](int q0, int q1) {
        for (int64_t q = q0; q < q1; ++q) {
          const int o = ba_pt_obs[(size_t)q];
          pt_uvw[3 * q] = ba_obs_uvw[3 * (size_t)o]; pt_uvw[3 * q + 1] = ba_obs_uvw[3 * (size_t)o + 1]; pt_uvw[3 * q + 2] = ba_obs_uvw[3 * (size_t)o + 2];
          pt_cam[(size_t)q] = ba_obs_cam[(size_t)o];
        }
      });
      HIPCHK(c, ba.d_pt_uvw.upload(pt_uvw, s)); HIPCHK(c, ba.d_pt_cam.upload(pt_cam, s));
      // ... and the landmarks' unary priors (PriorFactor<Point3>: mean, information in the upper-left 3x3 of the padded block)
      // packed in their order: the generic prior arrays are stored by variable, a gather of nine lines per landmark
      std::vector<int64_t> lp_ptr((size_t)n_lm + 1, 0);
      const int64_t NPall = (int64_t)c->prior_v.size();
      for (int64_t q = 0; q < NPall; ++q) { const int p = lm_index[c->prior_v[q]]; if (p >= 0) lp_ptr[(size_t)p + 1]++; }
      for (int p = 0; p < n_lm; ++p) lp_ptr[(size_t)p + 1] += lp_ptr[(size_t)p];
      std::vector<double> lp_val(9 * (size_t)lp_ptr[(size_t)n_lm]);
      {
        std::vector<int64_t> fill(lp_ptr.begin(), lp_ptr.end() - 1);
        for (int64_t q = 0; q < NPall; ++q) {                     // (ascending q: the order of the generic list of the variable)
          const int p = lm_index[c->prior_v[q]];
          if (p < 0) continue;
          double *o = &lp_val[9 * (size_t)fill[(size_t)p]++];
          const double *m = &c->prior_mean[(size_t)q * 7], *w = &c->prior_info[(size_t)q * 21];
          o[0] = m[0]; o[1] = m[1]; o[2] = m[2]; o[3] = w[0]; o[4] = w[1]; o[5] = w[2]; o[6] = w[6]; o[7] = w[7]; o[8] = w[11];
        }
      }
      HIPCHK(c, ba.d_lp_ptr.upload(lp_ptr, s)); HIPCHK(c, ba.d_lp_val.upload(lp_val, s));
      HIPCHK(c, hipStreamSynchronize(s));
      HIPCHK(c, ba.d_obs_cam.upload(ba_obs_cam, s)); HIPCHK(c, ba.d_obs_col.upload(ba_obs_col, s));
      HIPCHK(c, ba.d_obs_lm.upload(ba_obs_lm, s)); HIPCHK(c, ba.d_cam_ptr.upload(ba_cam_ptr, s)); HIPCHK(c, ba.d_cam_col.upload(ba_cam_col, s));
      HIPCHK(c, ba.d_tgt_blk.upload(ba_tgt_blk, s)); HIPCHK(c, ba.d_tgt_ptr.upload(ba_tgt_ptr, s));
      HIPCHK(c, ba.d_op_a.upload(ba_op_a, s)); HIPCHK(c, ba.d_op_b.upload(ba_op_b, s)); HIPCHK(c, ba.d_op_lm.upload(ba_op_lm, s));
      for (int i = 0; i < 2; ++i) {
        HIPCHK(c, ba.d_W[i].alloc(n_obs * 18)); HIPCHK(c, ba.d_Hpp[i].alloc((size_t)n_lm * 6)); HIPCHK(c, ba.d_bp[i].alloc((size_t)n_lm * 3));
      }
      HIPCHK(c, ba.d_Hinv.alloc((size_t)n_lm * 6)); HIPCHK(c, ba.d_zp.alloc((size_t)n_lm * 3)); HIPCHK(c, ba.d_pt_val.alloc((size_t)n_lm * 3));
      HIPCHK(c, ba.d_Hred.alloc(hblocks * 36)); HIPCHK(c, ba.d_bred.alloc((size_t)nb * 6));
      HIPCHK(c, hipStreamSynchronize(s));                  // the staging vectors die with this function
      HIPCHK(c, ba.d_lm_mine.upload(lm_mine, s));
      HIPCHK(c, hipStreamSynchronize(s));
      B.lm_mine = lm_mine.empty() ? nullptr : ba.d_lm_mine.p;
      B.n_lm = n_lm; B.n_obs = (int64_t)n_obs; B.n_tgt = (int)ba_tgt_blk.size(); B.n_cam = (int)ba_cam_col.size();
      B.lm_var = ba.d_lm_var.p; B.pt_ptr = ba.d_pt_ptr.p; B.pt_obs = ba.d_pt_obs.p; B.obs_uvw = ba.d_obs_uvw.p; B.obs_cam = ba.d_obs_cam.p; B.pt_uvw = ba.d_pt_uvw.p; B.pt_cam = ba.d_pt_cam.p; B.lp_ptr = ba.d_lp_ptr.p; B.lp_val = ba.d_lp_val.p;
      B.o_first = ba_o_first; B.n_tgt_small = ba_n_small; B.tgt_list = ba.d_tgt_list.p; B.n_tgt_list = (int)ba_tgt_list.size();
      B.cam_t0 = ba.d_cam_t0.p; B.cam_list = ba.d_cam_list.p; B.n_cam_list = (int)ba_cam_list.size();
      B.obs_col = ba.d_obs_col.p; B.obs_lm = ba.d_obs_lm.p; B.cam_ptr = ba.d_cam_ptr.p; B.cam_col = ba.d_cam_col.p;
      B.tgt_blk = ba.d_tgt_blk.p; B.tgt_ptr = ba.d_tgt_ptr.p; B.op_a = ba.d_op_a.p; B.op_b = ba.d_op_b.p; B.op_lm = ba.d_op_lm.p;
      B.Hinv = ba.d_Hinv.p; B.zp = ba.d_zp.p; B.pt_val = ba.d_pt_val.p;
      if (c->cfg.verbose)
        std::fprintf(stderr, "[fgo] landmark elimination: %d landmarks, %zu observations, %d reduced blocks with %zu landmark terms\n", n_lm, n_obs,
                     B.n_tgt, ba_op_a.size());
    }
  }
    return FGO_OK;
  }

  // ---- DevPlan / HostSchedule: the pointers and counts every kernel launch receives
  int fill_plan() {
    P.n_poses = NX; P.n_real = N; P.n_edges = E; P.edge_stride = E_cap; P.nb = nb;
    P.pose_col = c->d_pose_col.p; P.edge_i = c->d_edge_i.p; P.edge_j = c->d_edge_j.p;
    P.ainv = c->d_ainv.p; P.info = c->d_ainv.p + 8; P.edge_slot = c->d_edge_slot.p;
    P.he_ptr = c->d_he_ptr.p; P.he = c->d_he.p;
    P.hub_list = c->d_hub_list.p; P.hub_slice = c->d_hub_slice.p; P.n_hubs = (int)hubs.var.size(); P.hub_deg = hubs.deg_limit;
    P.hub_part = c->d_hub_part.p; P.hubm = c->d_hubm.p; P.n_hub_multi = (int)(hubs.multi.size() / 3);
    P.n_dup_groups = (int64_t)dup_ptr.size() - 1; P.dup_ptr = c->d_dup_ptr.p; P.dup_edges = c->d_dup_edges.p; P.dup_slot = c->d_dup_slot.p;
    P.n_priors = NP; P.prior_ptr = c->d_prior_ptr.p; P.prior_pose = c->d_prior_pose.p;
    P.prior_minv = c->d_prior_minv.p; P.prior_info = c->d_prior_info.p;
    P.var_kind = c->d_var_kind.p; P.edge_kind = c->d_edge_kind.p; P.cam = c->cam;
    P.n_imu = NI; P.imu = c->d_imu.p; P.imu_ids = c->d_imu_ids.p; P.imu_slot = c->d_imu_slot.p;
    P.imu_stash = c->d_imu_stash.p; P.imu_fn = (int64_t)imu_list.size(); P.imu_list = c->d_imu_list.p;
    P.imu_ncolor = (int)c->imu_color_ptr.size() - 1; P.imu_color_ptr_h = c->imu_color_ptr.data();
    for (int k = 0; k < 3; ++k) P.gravity[k] = c->gravity[k];
    P.n_hblocks = (int64_t)hblocks;
    P.lin_priors = 1;                               // (the prior CSR above already holds this rank's priors only)
    P.zero_offdiag = dist ? 1 : 0;
    P.dist = dist ? 1 : 0; P.top_col0 = top_col0; P.top_blk0 = top_blk0;
    P.top_ext0 = c->d_top_ext0.p; P.own_op0 = c->d_own_op0.p; P.own_op1 = c->d_own_op1.p;
    P.top_row0 = c->d_top_row0.p; P.own_row0 = c->d_own_row0.p; P.own_row1 = c->d_own_row1.p;
    P.var_mine = dist ? c->d_var_mine.p : nullptr; P.lambda_rank = rank == 0 ? 1 : 0;
    c->sched.world = world; c->sched.rank = rank; c->sched.seg_group = S.seg_group;
    c->sched.n_top_blocks = S.nnzL - top_blk0; c->sched.n_top_cols = nb - top_col0;
    P.colptr = c->d_colptr.p; P.rowidx = c->d_rowidx.p; P.asrc = c->d_asrc.p;
    P.zero_blk = (int)S.nnzL;
    P.prof_tri = std::getenv("FGO_TRI_PROF") ? std::atoi(std::getenv("FGO_TRI_PROF")) : 0;
    P.op_ptr = c->d_op_ptr.p; P.op_mid = c->d_op_mid.p; P.op_a = c->d_op_a.p; P.op_b = c->d_op_b.p;
    P.acc_targets = c->d_acc_targets.p;
    P.g2_tgt = c->d_g2_tgt.p; P.g2_ptr = c->d_g2_ptr.p; P.g2_b = c->d_g2_b.p; P.g2_a = c->d_g2_a.p;
    P.task_dirty = nullptr; P.acc_task = c->d_acc_task.p; P.g2_task = c->d_g2_task.p; P.tcol_task = c->d_tcol_task.p;
    P.ride_xcd = (int)tune("ride_xcd", 1);
    P.ride_items = S.ride_items.empty() ? nullptr : c->d_ride_items.p; P.acc_start = S.acc_start.empty() ? nullptr : c->d_acc_start.p; P.acc_desc = c->d_acc_desc.p;
    c->sched.ride_ptr = S.ride_items.empty() ? std::vector<int>() : S.ride_ptr;
    c->sched.g2_lvl = S.g2_lvl;
    if (S.g2_ptr.size() <= 1) c->sched.g2_lvl.clear();
    P.rowptr = c->d_rowptr.p; P.row_blk = c->d_row_blk.p; P.row_col = c->d_row_col.p;
    P.task_ptr = c->d_task_ptr.p; P.task_cols = c->d_task_cols.p;
    P.partial = c->d_partial.p;
    P.pp.task_panel = c->d_task_panel.p; P.pp.panel_task = c->d_panel_task.p; P.pp.ptri_blk = c->d_ptri_blk.p;
    P.pp.prow_ptr = c->d_prow_ptr.p; P.pp.prow_idx = c->d_prow_idx.p; P.pp.prow_blk = c->d_prow_blk.p;
    P.pp.pchunk_panel = c->d_pchunk_panel.p; P.pp.pchunk_row0 = c->d_pchunk_row0.p; P.pp.pchunk_nrows = c->d_pchunk_nrows.p;
    P.pp.panel_chunk0 = c->d_panel_chunk0.p; P.pp.row_mid = c->d_row_mid.p; P.pp.fchunk_col = c->d_fchunk_col.p;
    P.pp.fchunk_e0 = c->d_fchunk_e0.p; P.pp.pcol_fchunk0 = c->d_pcol_fchunk0.p; P.pp.pcol_fchunkn = c->d_pcol_fchunkn.p;
    P.pp.fpart = c->d_fpart.p; P.pp.bpart = c->d_bpart.p; P.pp.ptop = c->d_ptop.p;
    P.pp.rchunk_panel = c->d_rchunk_panel.p; P.pp.rchunk_s0 = c->d_rchunk_s0.p; P.pp.rchunk_src = c->d_rchunk_src.p; P.pp.rchunk_src0 = byc_c0; c->sched.rows_byc_level = c->d_rchunk_src.p ? byc_level : (1 << 30);
    P.pp.ptri_src = c->d_ptri_src.p; P.pp.prow_src = c->d_prow_src.p;
    P.pp.pdesc = c->d_pdesc.p; P.pp.tri_order = c->d_tri_order.p; P.pp.leaf_desc = c->d_leaf_desc.p; P.pp.leaf_lpt = c->d_leaf_lpt.p; P.pp.rchunks = c->d_rchunks.p; P.pp.bchunks = c->d_bchunks.p; P.pp.bchain = c->d_bchain.p; P.pp.bchain_done = c->d_bchain_done.p;
    c->sched.level_panel = S.level_panel; c->sched.pchunk_ptr = S.pchunk_ptr; c->sched.fchunk_ptr = S.fchunk_ptr; c->sched.rchunk_ptr = S.rchunk_ptr;
    if (std::getenv("FGO_NO_PANELS")) std::fill(c->sched.level_panel.begin(), c->sched.level_panel.end(), 0);
    c->sched.level_leaf = S.level_leaf; c->sched.level_leaf_maxblk = S.level_leaf_maxblk; c->sched.level_leaf_maxops = S.level_leaf_maxops;
    if (tune("no_leaf", 0) != 0) std::fill(c->sched.level_leaf.begin(), c->sched.level_leaf.end(), 0);
    if (prof)
      for (size_t l = 0; l < S.level_leaf.size(); ++l)
        if (S.level_leaf[l]) {
          // distribution of the leaf tasks' sizes: the level's launch reserves LDS for the LARGEST task in every workgroup
          std::vector<int64_t> nb_t;
          int64_t cols = 0;
          for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) {
            const int c0 = S.task_ptr[t], m = S.task_ptr[t + 1] - c0;
            nb_t.push_back(S.colptr[S.task_cols[c0] + m] - S.colptr[S.task_cols[c0]]); cols += m;
          }
          std::sort(nb_t.begin(), nb_t.end());
          std::fprintf(stderr, "[fgo build]    leaf level %zu: %zu tasks, %lld columns, blocks per task min %lld / median %lld / p90 %lld / max %d, most ops %d\n", l,
                       nb_t.size(), (long long)cols, (long long)nb_t.front(), (long long)nb_t[nb_t.size() / 2], (long long)nb_t[nb_t.size() * 9 / 10], S.level_leaf_maxblk[l],
                       S.level_leaf_maxops[l]);
        }
    c->sched.n_levels = (int)S.level_ptr.size() - 1;
    c->sched.level_ptr = S.level_ptr;
    c->sched.acc_ptr = S.acc_ptr; c->sched.acc_mid = S.acc_mid;
    c->sched.level_pn0.assign(c->sched.n_levels, 0);
    for (int l = 0; l < c->sched.n_levels; ++l)
      if (S.level_panel[l]) {
        c->sched.level_pn0[l] = S.task_panel[S.level_ptr[l]];
        for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t)
          if (S.task_panel[t] != c->sched.level_pn0[l] + (t - S.level_ptr[l])) return fail(c, FGO_EINVAL, "internal: panel ids of a level are not consecutive");
      }
    c->sched.level_col_ptr.resize(c->sched.n_levels + 1);
    for (int l = 0; l <= c->sched.n_levels; ++l) c->sched.level_col_ptr[l] = S.task_ptr[S.level_ptr[l]];
    {
      // forward-solve work items: one per panel column, or one per chunk of FWD_CHUNK entries where the external part of the
      // column's row is longer than that (not in distributed mode: a top row's domain part arrives by collective)
      static const int fwd_split = (int)tune("fwd_split", 32);     // least number of chunks of FWD_CHUNK entries (0: never split); a level with split rows pays one more (tiny) launch
      std::vector<int> fwg_ci, fwg_ch, fsplit, f0v((size_t)nb, 0), fnv((size_t)nb, 1);
      c->sched.fwg_ptr.assign((size_t)c->sched.n_levels + 1, 0);
      c->sched.fsplit_ptr.assign((size_t)c->sched.n_levels + 1, 0);
      for (int l = 0; l < c->sched.n_levels; ++l) {
        if (S.level_panel[l])
          for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) {
            const int pn = S.task_panel[t], m = S.task_ptr[t + 1] - S.task_ptr[t];
            for (int q = 0; q < m; ++q) {
              const int ci = S.task_ptr[t] + q;
              const int f0 = S.pcol_fchunk0[S.col_off(pn) + q], fn = S.pcol_fchunkn[S.col_off(pn) + q];
              if (fn <= 1 || dist || !fwd_split || fn < fwd_split) { fwg_ci.push_back(ci); fwg_ch.push_back(-1); continue; }
              f0v[(size_t)ci] = f0; fnv[(size_t)ci] = fn; fsplit.push_back(ci);
              for (int x = 0; x < fn; ++x) { fwg_ci.push_back(ci); fwg_ch.push_back(f0 + x); }
            }
          }
        c->sched.fwg_ptr[(size_t)l + 1] = (int)fwg_ci.size();
        c->sched.fsplit_ptr[(size_t)l + 1] = (int)fsplit.size();
      }
      HIPCHK(c, c->d_fwg_ci.upload(fwg_ci, s)); HIPCHK(c, c->d_fwg_ch.upload(fwg_ch, s));
      HIPCHK(c, c->d_fwd_f0.upload(f0v, s)); HIPCHK(c, c->d_fwd_fn.upload(fnv, s));
      HIPCHK(c, c->d_fsplit_ci.upload(fsplit, s));
      HIPCHK(c, hipStreamSynchronize(s));
      P.fwg_ci = c->d_fwg_ci.p; P.fwg_ch = c->d_fwg_ch.p; P.fwd_f0 = c->d_fwd_f0.p; P.fwd_fn = c->d_fwd_fn.p; P.fsplit_ci = c->d_fsplit_ci.p;
  }
  c->sched.level_maxcol.assign(c->sched.n_levels, 0);
  c->sched.level_maxrow.assign(c->sched.n_levels, 0);
  c->sched.level_maxtaskcols.assign(c->sched.n_levels, 0);
  for (int l = 0; l < c->sched.n_levels; ++l)
    for (int t = S.level_ptr[l]; t < S.level_ptr[l + 1]; ++t) {
      c->sched.level_maxtaskcols[l] = std::max(c->sched.level_maxtaskcols[l], S.task_ptr[t + 1] - S.task_ptr[t]);
      for (int q = S.task_ptr[t]; q < S.task_ptr[t + 1]; ++q) {
        const int k = S.task_cols[q];
        c->sched.level_maxcol[l] = std::max(c->sched.level_maxcol[l], (int)(S.colptr[k + 1] - S.colptr[k]));
        c->sched.level_maxrow[l] = std::max(c->sched.level_maxrow[l], (int)(S.rowptr[k + 1] - S.rowptr[k]));
      }
    }
  if (R > 0) { c->inc.E_cap = E_cap; c->inc.NI_cap = NI_cap; c->inc.valid = true; }
  c->built_N = N;
  c->cur = 0;
  c->cov_factor_valid = false;
  c->h_pose_col.clear();
  c->structure_dirty = false;
  c->host_poses_newer = true;
  c->lin_valid = false;

    return FGO_OK;
  }

  // ---- incremental-mode bookkeeping, statistics
  int finish() {
    fgo_stats &st = c->last;
    std::memset(&st, 0, sizeof(st));
    st.structure_rebuilt = 1;
    st.t_symbolic = t1 - t0;
    st.t_upload = now_s() - t1;
    st.n_free = nb - (int)R + n_lm; st.n_edges = E;         // the phantom slots of the incremental mode are not the caller's variables
    st.nnz_H_blocks = (int64_t)hblocks; st.nnz_L_blocks = S.nnzL; st.n_update_ops = S.nops;
    st.n_levels = c->sched.n_levels; st.n_tasks = (int)S.task_ptr.size() - 1;
    // algorithmic HBM bytes: factor = read H once + write L once; solve = read L twice;
    // linearise = edge payload (232 B) + two 64-B pose gathers per edge, once per half-edge, + H/b written once
    // (the forward solve is fused into the factor sweep: it re-reads L once there; the solve phase is the backward sweep)
    st.bytes_factor = 288.0 * (double)hblocks + 2.0 * 288.0 * (double)S.nnzL + 2.0 * 48.0 * nb;
    st.bytes_solve = 288.0 * (double)S.nnzL + 2.0 * 48.0 * nb;
    st.bytes_linearize = (double)E * (8 + 56 + 168) + (double)E * 2 * 56 + 288.0 * (double)hblocks + 48.0 * nb;
    if (n_lm > 0) {
      // landmark elimination: pixel + weight (24 B), camera pose (64 B), landmark (32 B) per observation and side, the coupling
      // block W written once (144 B); per trial W read, Y written and read once more by the reduction (3 x 144 B) plus the
      // 3x3 blocks; the back-substitution reads Y again
      const double no = (double)ba_obs_uvw.size() / 3.0;
      st.bytes_linearize += no * (2.0 * (24 + 64 + 32) + 144) + 72.0 * n_lm;
      st.bytes_factor += no * 3.0 * 144 + 144.0 * n_lm;
      st.bytes_solve += no * 144 + 96.0 * n_lm;
  }
  if (c->cfg.verbose)
    std::fprintf(stderr, "[fgo] build: N=%lld E=%lld free=%d nnzL=%lld ops=%lld levels=%d tasks=%d symbolic %.3fs (ordering %.3fs) upload %.3fs\n",
                 (long long)N, (long long)E, nb, (long long)S.nnzL, (long long)S.nops, st.n_levels, st.n_tasks, st.t_symbolic, t_ord1 - t_ord0, st.t_upload);
  // host copies of the big lists are no longer needed.  Unmapping a few hundred MB takes tens of milliseconds (cfg 2: 31 ms
  // here + 30 ms for this object's own tables, a seventh of a fresh context's first optimize()): a detached thread does it.
  {
    struct Lists { IntList a, b, c, d; };
    Lists *g = new Lists;
    g->a.swap(S.op_a); g->b.swap(S.op_b); g->c.swap(S.g2_a); g->d.swap(S.g2_b);
    try { std::thread([g] { delete g; }).detach(); }
    catch (...) { delete g; }                              // (thread limit reached: release here, on the caller's time)
  }
  return FGO_OK;
  }
};
}  // namespace

static int build_phases(fgo_ctx *c);
int build(fgo_ctx *c) {
  const double t0 = now_s();
  const int rc = build_phases(c);
  if (std::getenv("FGO_SYM_PROFILE")) std::fprintf(stderr, "[fgo build]    build() %.1f ms in all (incl. releasing its host tables)\n", 1e3 * (now_s() - t0));
  return rc;
}
static int build_phases(fgo_ctx *c) {
  // (the object's host tables are released by a detached thread once everything is on the device, see finish())
  // (a deleter must not throw: when no thread can be created -- EAGAIN under a pids / ulimit cap -- the tables are released here)
  struct Release { void operator()(StructureBuild *p) const noexcept { try { std::thread([p] { delete p; }).detach(); } catch (...) { delete p; } } };
  std::unique_ptr<StructureBuild, Release> holder(new StructureBuild(c));
  StructureBuild &sb = *holder;
  int rc;
  if ((rc = sb.classify()) != FGO_OK) return rc;
  if ((rc = sb.pairs_and_graph()) != FGO_OK) return rc;
  if ((rc = sb.order_and_symbolic()) != FGO_OK) return rc;
  if ((rc = sb.host_tables()) != FGO_OK) return rc;
  if ((rc = sb.upload()) != FGO_OK) return rc;
  if ((rc = sb.fill_plan()) != FGO_OK) return rc;
  if ((rc = sb.finish()) != FGO_OK) return rc;
  // the poses go up HERE, while the host tables are still alive: their release (a detached thread unmapping ~1.5 GB) holds the
  // address-space lock, and the first thing after a build -- a fresh staging vector for the poses, its page faults, the pinning of the
  // copy -- used to wait 45-70 ms behind it (cfg 2; measured round 5)
  if (c->host_poses_newer && (rc = upload_poses(c)) != FGO_OK) return rc;
  if (std::getenv("FGO_SYM_PROFILE")) std::fprintf(stderr, "[fgo build]    phases done after %.1f ms (symbolic %.1f + upload %.1f)\n", 1e3 * (now_s() - sb.t_enter), 1e3 * c->last.t_symbolic, 1e3 * c->last.t_upload);
  return FGO_OK;
}

static int upload_hubs(fgo_ctx *c, const HubPlan &hp, size_t entry_cap) {
  hipStream_t s = c->stream;
  HIPCHK(c, c->d_hub_list.alloc(std::max(entry_cap, hp.var.size())));
  HIPCHK(c, c->d_hub_slice.alloc(std::max(entry_cap, hp.var.size())));
  HIPCHK(c, c->d_hub_part.alloc(std::max(entry_cap, hp.var.size()) * HUB_PART));
  HIPCHK(c, c->d_hubm.alloc(3 * std::max(entry_cap, hp.var.size())));
  if (!hp.var.empty()) {
    HIPCHK(c, hipMemcpyAsync(c->d_hub_list.p, hp.var.data(), sizeof(int) * hp.var.size(), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_hub_slice.p, hp.slice.data(), sizeof(int) * hp.slice.size(), hipMemcpyHostToDevice, s));
  }
  if (!hp.multi.empty()) HIPCHK(c, hipMemcpyAsync(c->d_hubm.p, hp.multi.data(), sizeof(int) * hp.multi.size(), hipMemcpyHostToDevice, s));
  HIPCHK(c, hipStreamSynchronize(s));        // the caller's HubPlan may die right after
  return FGO_OK;
}

// Incremental mode: the graph grew since the structure was built.  If the new variables fit the phantom slots and every
// new factor couples variables whose pair already exists in the structure, the factor-side device arrays are extended
// in place: returns FGO_OK (done), 1 (does not fit: the caller rebuilds), or an error.
int refresh_factors(fgo_ctx *c) {
  fgo_ctx::Incr &I = c->inc;
  const bool growing = c->gtsam_mode ? c->isam_incremental : c->grow_incremental;
  if (!I.valid || !growing || c->shard_world > 1) return 1;
  const double t0 = now_s();
  const int64_t N = (int64_t)c->ids.size(), E = (int64_t)c->ei.size(), NI = (int64_t)c->imu_payload.size();
  if (N > I.NX || E > I.E_cap || NI > I.NI_cap || N < I.N_done || E < I.E_done || NI < I.NI_done) return 1;
  for (int64_t v = I.N_done; v < N; ++v) if (c->fixed[v]) return 1;
  // (a context holds either g2o-semantics edges or GTSAM-semantics factors: anything that would change the mode rebuilds, and the
  //  build reports it)
  for (int64_t e = I.E_done; e < E; ++e) if ((c->torder[e] == FGO_TANGENT_G2O) == c->gtsam_mode || (c->torder[e] == 3 && !c->cam_set)) return 1;
  if (!c->gtsam_mode) {
    if (!c->prior_v.empty() || NI > 0) return 1;
    for (int64_t v = I.N_done; v < N; ++v) if (c->var_kind[v] != 0) return 1;
  }
  auto find_pair = [&](int va, int vb) -> int {            // variable indices -> pair index, -1 none needed, -2 missing
    const int a = I.hidx[va], b = I.hidx[vb];
    if (a < 0 || b < 0 || a == b) return -1;
    const uint64_t key = ((uint64_t)(uint32_t)std::min(a, b) << 32) | (uint32_t)std::max(a, b);
    auto it = std::lower_bound(I.ukey.begin(), I.ukey.end(), key);
    return (it != I.ukey.end() && *it == key) ? (int)(it - I.ukey.begin()) : -2;
  };
  // ---- check everything first: nothing is modified unless the whole delta fits
  std::vector<int> new_h((size_t)(E - I.E_done));
  for (int64_t e = I.E_done; e < E; ++e) { const int h = find_pair(c->ei[e], c->ej[e]); if (h == -2) return 1; new_h[(size_t)(e - I.E_done)] = h; }
  std::vector<int> new_imu_slot((size_t)15 * (NI - I.NI_done), -1);
  for (int64_t f = I.NI_done; f < NI; ++f) {
    int q = 0;
    for (int u = 0; u < 6; ++u)
      for (int w = u + 1; w < 6; ++w, ++q) {
        const int vu = c->imu_ids[6 * f + u], vw = c->imu_ids[6 * f + w];
        const int h = find_pair(vu, vw);
        if (h == -2) return 1;
        if (h >= 0) new_imu_slot[(size_t)15 * (f - I.NI_done) + q] = (int)((((int64_t)I.nb + h) << 1) | (I.pose_col[vw] > I.pose_col[vu] ? 1 : 0));
      }
  }
  {   // hub entries (one workgroup per slice of a hub variable) after the extension: the scratch buffers have room for hub_cap
    std::unordered_map<int, int64_t> deg;
    for (int64_t e = I.E_done; e < E; ++e)
      for (const int v : {c->ei[e], c->ej[e]}) {
        auto it = deg.find(v);
        if (it == deg.end()) it = deg.emplace(v, I.he_ptr[v + 1] - I.he_ptr[v]).first;
        ++it->second;
      }
    auto entries = [&](int64_t d) -> int64_t { return d > I.hub_deg ? std::min<int64_t>(HUB_MAX_SLICES, (d + HUB_SLICE - 1) / HUB_SLICE) : 0; };
    int64_t n_entries = (int64_t)c->plan.n_hubs;
    for (auto &kv : deg) n_entries += entries(kv.second) - entries(I.he_ptr[kv.first + 1] - I.he_ptr[kv.first]);
    if (n_entries > (int64_t)I.hub_cap) return 1;
  }
  if (c->dev_poses_newer) { const int rc = download_poses(c); if (rc) return rc; }
  destroy_graphs(c);                                           // captured trials hold the factor counts by value
  hipStream_t s = c->stream;
  I.valid = false;                                             // (an error below leaves the lists half-extended: the next call rebuilds)
  // ---- variables: claim phantom slots (kind; the value goes up with upload_poses)
  if (N > I.N_done) {
    HIPCHK(c, hipMemcpyAsync(c->d_var_kind.p + I.N_done, c->var_kind.data() + I.N_done, sizeof(int) * (size_t)(N - I.N_done), hipMemcpyHostToDevice, s));
    c->host_poses_newer = true;
  }
  // ---- binary factors: slots, duplicate groups, payload, incidence lists
  I.edge_h.resize((size_t)E, -1); I.edge_slot.resize((size_t)E, -1);
  int64_t slot_lo = E;                                          // lowest edge whose slot entry changed
  for (int64_t e = I.E_done; e < E; ++e) {
    const int h = new_h[(size_t)(e - I.E_done)];
    I.edge_h[e] = h;
    if (h < 0) continue;
    const int slot = (int)((((int64_t)I.nb + h) << 1) | (I.pose_col[c->ej[e]] > I.pose_col[c->ei[e]] ? 1 : 0));
    if (I.pair_nbin[h] == 0) { I.pair_first[h] = (int)e; I.edge_slot[e] = slot; }
    else {
      if (I.pair_nbin[h] == 1) { const int f0 = I.pair_first[h]; I.dups[h].push_back(f0); I.edge_slot[f0] = -1; slot_lo = std::min<int64_t>(slot_lo, f0); }
      I.dups[h].push_back(e);
    }
    ++I.pair_nbin[h];
  }
  slot_lo = std::min(slot_lo, I.E_done);
  if (E > slot_lo) HIPCHK(c, hipMemcpyAsync(c->d_edge_slot.p + slot_lo, I.edge_slot.data() + slot_lo, sizeof(int) * (size_t)(E - slot_lo), hipMemcpyHostToDevice, s));
  std::vector<int64_t> dup_ptr{0}, dup_edges;
  std::vector<int> dup_slot;
  for (auto &kv : I.dups) {
    for (int64_t e : kv.second) {
      dup_edges.push_back(e);
      dup_slot.push_back((int)((((int64_t)I.nb + kv.first) << 1) | (I.pose_col[c->ej[e]] > I.pose_col[c->ei[e]] ? 1 : 0)));
    }
    dup_ptr.push_back((int64_t)dup_edges.size());
  }
  HIPCHK(c, c->d_dup_ptr.upload(dup_ptr, s));
  HIPCHK(c, c->d_dup_edges.upload(dup_edges, s));
  HIPCHK(c, c->d_dup_slot.upload(dup_slot, s));
  const int64_t dE = E - I.E_done;
  std::vector<double> stage((size_t)28 * dE);
  if (dE > 0) {
    HIPCHK(c, hipMemcpyAsync(c->d_edge_i.p + I.E_done, c->ei.data() + I.E_done, sizeof(int) * (size_t)dE, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_edge_j.p + I.E_done, c->ej.data() + I.E_done, sizeof(int) * (size_t)dE, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_edge_kind.p + I.E_done, c->torder.data() + I.E_done, sizeof(int) * (size_t)dE, hipMemcpyHostToDevice, s));
    for (int64_t e = I.E_done; e < E; ++e) {                    // SoA payload: staged contiguously, scattered by a kernel
      double *o = &stage[(size_t)28 * (e - I.E_done)];
      if (c->torder[e] <= 1) pose_inv7(&c->meas[(size_t)e * 7], o); else std::memcpy(o, &c->meas[(size_t)e * 7], 7 * sizeof(double));
      std::memcpy(o + 7, &c->info[(size_t)e * 21], 21 * sizeof(double));
    }
    HIPCHK(c, c->d_stage.alloc(stage.size()));
    HIPCHK(c, hipMemcpyAsync(c->d_stage.p, stage.data(), sizeof(double) * stage.size(), hipMemcpyHostToDevice, s));
    launch_scatter_edges(c->d_stage.p, dE, I.E_done, c->d_ainv.p, s);
  }
  // incidence lists: a new factor's half-edges go to the END of its variables' lists (edge order inside a list is what keeps
  // the sums deterministic); everything behind the lowest touched variable moves up and is uploaded again -- new factors
  // attach to the newest variables, so that is a short suffix
  {
    // one merge pass: per-variable insert counts -> new offsets by a prefix sum -> the old lists copied in order with the
    // new half-edges appended behind each variable's old ones (in edge order): O(NX + moved suffix), however many factors
    int64_t v_lo = I.NX;
    if (E > I.E_done) {
      std::vector<std::pair<int, int>> ins;                      // (variable, half-edge), edge order
      ins.reserve((size_t)2 * (E - I.E_done));
      for (int64_t e = I.E_done; e < E; ++e) {
        ins.push_back({c->ei[e], (int)(e << 1)});
        ins.push_back({c->ej[e], (int)((e << 1) | 1)});
        v_lo = std::min<int64_t>(v_lo, std::min(c->ei[e], c->ej[e]));
      }
      std::stable_sort(ins.begin(), ins.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
      const int64_t p_lo = I.he_ptr[v_lo];
      std::vector<int> tail(I.he.begin() + p_lo, I.he.end());   // old entries of the variables >= v_lo
      I.he.resize(I.he.size() + ins.size());
      size_t q = 0;
      int64_t w = p_lo, shift = 0;
      for (int64_t v = v_lo; v < I.NX; ++v) {
        const int64_t o0 = I.he_ptr[v] - shift - p_lo, o1 = I.he_ptr[v + 1] - p_lo;   // he_ptr[v] already carries `shift`, he_ptr[v+1] not yet
        for (int64_t r = o0; r < o1; ++r) I.he[w++] = tail[(size_t)r];
        while (q < ins.size() && ins[q].first == v) { I.he[w++] = ins[q].second; ++q; ++shift; }
        I.he_ptr[v + 1] += shift;
      }
    }
    if (v_lo < I.NX) {
      const int64_t p0 = I.he_ptr[v_lo];
      HIPCHK(c, hipMemcpyAsync(c->d_he.p + p0, I.he.data() + p0, sizeof(int) * (size_t)((int64_t)I.he.size() - p0), hipMemcpyHostToDevice, s));
      HIPCHK(c, hipMemcpyAsync(c->d_he_ptr.p + v_lo, I.he_ptr.data() + v_lo, sizeof(int64_t) * (size_t)(I.NX + 1 - v_lo), hipMemcpyHostToDevice, s));
    }
    HubPlan hubs;
    plan_hubs(I.he_ptr, I.NX, I.hub_deg, hubs);
    { const int rc = upload_hubs(c, hubs, I.hub_cap); if (rc) return rc; }
    c->plan.n_hubs = (int)hubs.var.size(); c->plan.n_hub_multi = (int)(hubs.multi.size() / 3);
  }
  // ---- priors (few): rebuilt
  if ((int64_t)c->prior_v.size() != I.NP_done) {
    std::vector<unsigned char> all((size_t)I.NX, 1);
    const int rc = upload_priors(c, I.NX, all);
    if (rc) return rc;
  }
  // ---- IMU factors: payload / ids / slots appended, incidence rebuilt
  const int64_t dI = NI - I.NI_done;
  std::vector<int> imu_sorted;
  if (dI > 0) {
    HIPCHK(c, hipMemcpyAsync(c->d_imu.p + I.NI_done, c->imu_payload.data() + I.NI_done, sizeof(ImuPayload) * (size_t)dI, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_imu_ids.p + 6 * I.NI_done, c->imu_ids.data() + 6 * I.NI_done, sizeof(int) * (size_t)(6 * dI), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_imu_slot.p + 15 * I.NI_done, new_imu_slot.data(), sizeof(int) * (size_t)(15 * dI), hipMemcpyHostToDevice, s));
  }
  if (dI > 0) {
    int64_t v_lo = I.NX;
    {
      std::vector<std::pair<int, int>> ins;                      // (variable, incidence entry), factor order
      for (int64_t f = I.NI_done; f < NI; ++f)
        for (int u = 0; u < 6; ++u) {
          const int v = c->imu_ids[6 * f + u];
          ins.push_back({v, (int)((f << 3) | u)});
          v_lo = std::min<int64_t>(v_lo, v);
        }
      std::stable_sort(ins.begin(), ins.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
      const int64_t p_lo = I.imu_inc_ptr[v_lo];
      std::vector<int> tail(I.imu_inc.begin() + p_lo, I.imu_inc.end());
      I.imu_inc.resize(I.imu_inc.size() + ins.size());
      size_t q = 0;
      int64_t w = p_lo, shift = 0;
      for (int64_t v = v_lo; v < I.NX; ++v) {
        const int64_t o0 = I.imu_inc_ptr[v] - shift - p_lo, o1 = I.imu_inc_ptr[v + 1] - p_lo;
        for (int64_t r = o0; r < o1; ++r) I.imu_inc[w++] = tail[(size_t)r];
        while (q < ins.size() && ins[q].first == v) { I.imu_inc[w++] = ins[q].second; ++q; ++shift; }
        I.imu_inc_ptr[v + 1] += shift;
      }
    }
    // the new factors' colours; the colour-sorted list again
    for (int64_t f = I.NI_done; f < NI; ++f) imu_colour_add(c, f);
    imu_colour_lists(c, imu_sorted);
    HIPCHK(c, hipMemcpyAsync(c->d_imu_list.p, imu_sorted.data(), sizeof(int) * imu_sorted.size(), hipMemcpyHostToDevice, s));
  }
  HIPCHK(c, hipStreamSynchronize(s));                           // the staging vectors die here
  // ---- plan
  DevPlan &P = c->plan;
  P.n_edges = E; P.n_real = N; P.hub_list = c->d_hub_list.p; P.hub_slice = c->d_hub_slice.p; P.hubm = c->d_hubm.p; P.hub_part = c->d_hub_part.p;
  P.he_ptr = c->d_he_ptr.p; P.he = c->d_he.p;
  P.n_dup_groups = (int64_t)dup_ptr.size() - 1; P.dup_ptr = c->d_dup_ptr.p; P.dup_edges = c->d_dup_edges.p; P.dup_slot = c->d_dup_slot.p;
  P.n_priors = c->n_priors_dev; P.prior_ptr = c->d_prior_ptr.p; P.prior_pose = c->d_prior_pose.p;
  P.prior_minv = c->d_prior_minv.p; P.prior_info = c->d_prior_info.p;
  P.n_imu = NI; P.imu_fn = (int64_t)c->imu_flist.size();
  P.imu_ncolor = (int)c->imu_color_ptr.size() - 1; P.imu_color_ptr_h = c->imu_color_ptr.data();
  for (int k = 0; k < 3; ++k) P.gravity[k] = c->gravity[k];
  P.cam = c->cam;
  I.N_done = N; I.E_done = E; I.NI_done = NI; I.NP_done = (int64_t)c->prior_v.size();
  c->n_phantom = (int)(I.NX - N);                               // the new variables took the first phantom slots: the dense read-backs report them
  I.valid = true;
  c->structure_dirty = false;
  c->lin_valid = false;
  c->cov_factor_valid = false;
  fgo_stats &st = c->last;
  st.structure_rebuilt = 0;
  st.t_symbolic = now_s() - t0;                                 // host time of the in-place extension
  st.t_upload = 0;
  st.n_edges = E;
  st.n_free = I.nb - c->n_phantom;
  return FGO_OK;
}

}  // namespace fgo
