// Internal host-side data structures of libfgo (not part of the C-ABI).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <thread>
#include <pthread.h>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <atomic>
#include <algorithm>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace fgo {

// std::vector whose resize() leaves new elements uninitialised: the big index lists (hundreds of MB at cfg 5) are
// written exactly once by several host threads, a zero-filling resize would touch every page serially first
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { typedef NoInitAlloc<U> other; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
  template <class U> void construct(U *p) { ::new ((void *)p) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
typedef std::vector<int, NoInitAlloc<int>> IntList;

// fn(begin, end) over [0, n) in chunks handed out dynamically to a few host threads (FGO_HOST_THREADS, default
// min(hardware threads, 32)).  Callers write disjoint outputs per index, so results do not depend on the thread count.
// The workers are a process-wide pool created on first use (the structure phase makes dozens of these calls; spawning
// and joining the threads each time cost about a millisecond per call); one parallel region at a time, nested or
// concurrent callers (two contexts built from two host threads) simply run their region on the calling thread.
inline int host_threads() {
  static const int nthreads = [] {
    const char *e = std::getenv("FGO_HOST_THREADS");
    int t = e ? std::atoi(e) : (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    return std::max(1, t);
  }();
  return nthreads;
}
// ---- tuning.  The launch-selection thresholds, the capacity model of the riders, the ordering's balance terms ... are
// constants next to the code that uses them (with the measurement that set them); ONE environment variable overrides any of
// them for a sweep or an A/B run:   FGO_TUNE="ride_win=25,acc2_min=6000,tri1_min=0"   (parsed once per process).
// Functional switches that tests flip per context keep their own variables (FGO_BA_SCHUR, FGO_TASK_WORK, FGO_NO_PANELS, ...).
inline double tune(const char *key, double dflt) {
  static const std::unordered_map<std::string, double> *tab = [] {
    auto *m = new std::unordered_map<std::string, double>();
    if (const char *e = std::getenv("FGO_TUNE")) {
      std::string s(e);
      size_t i = 0;
      while (i < s.size()) {
        size_t j = s.find(',', i);
        if (j == std::string::npos) j = s.size();
        const std::string kv = s.substr(i, j - i);
        const size_t q = kv.find('=');
        if (q != std::string::npos && q > 0) (*m)[kv.substr(0, q)] = std::atof(kv.c_str() + q + 1);
        i = j + 1;
      }
    }
    return m;
  }();
  const auto it = tab->find(key);
  return it == tab->end() ? dflt : it->second;
}
// The launch selection is expressed in multiples of the compute units of the device a context lives on -- how many panels make
// a level "wide", how many idle CUs a rider slot has -- not in absolute numbers.  The count is a property of the CONTEXT
// (Symbolic::cus / HostSchedule::cus, set by build()): two contexts on different devices may build at the same time.
class HostPool {
 public:
  // (never destroyed: worker threads end with the process; a forked child starts with no pool of its own and creates a new one)
  static HostPool &get() {
    static std::once_flag once;
    std::call_once(once, [] { pthread_atfork(nullptr, nullptr, [] { instance().store(nullptr); }); });
    HostPool *p = instance().load();
    if (!p) {
      HostPool *fresh = new HostPool;
      if (instance().compare_exchange_strong(p, fresh)) p = fresh; else delete fresh;
    }
    return *p;
  }
  // runs job(worker index) on every pool thread and on the caller; returns when all are done
  template <class F>
  bool run(int nworkers, F &&job) {
    std::unique_lock<std::mutex> own(busy_, std::try_to_lock);
    if (!own.owns_lock()) return false;                       // pool in use (nested / concurrent region): caller runs alone
    ensure(nworkers);
    // the bodies allocate (push_back, vector construction): an exception on ANY thread is parked, every worker is waited
    // for (they hold references to the caller's stack), then it is rethrown on the caller -- never std::terminate, never a
    // caller unwinding under running workers
    std::exception_ptr err;
    std::mutex err_m;
    auto guarded = [&](int w) {
      try { job(w); }
      catch (...) { std::lock_guard<std::mutex> g(err_m); if (!err) err = std::current_exception(); }
    };
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = guarded;
      active_ = std::min(nworkers, (int)th_.size());
      pending_ = active_;
      ++epoch_;
    }
    cv_.notify_all();
    guarded(-1);
    {
      std::unique_lock<std::mutex> lk(m_);
      done_.wait(lk, [&] { return pending_ == 0; });
      job_ = nullptr;
    }
    if (err) std::rethrow_exception(err);
    return true;
  }
 private:
  static std::atomic<HostPool *> &instance() { static std::atomic<HostPool *> p{nullptr}; return p; }
  void ensure(int n) {
    while ((int)th_.size() < n) {
      const int id = (int)th_.size();
      uint64_t seen;
      { std::lock_guard<std::mutex> lk(m_); seen = epoch_; }
      th_.emplace_back([this, id, seen]() mutable {
        for (;;) {
          std::function<void(int)> job;
          {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return epoch_ != seen; });
            seen = epoch_;
            if (id >= active_) continue;
            job = job_;
          }
          job(id);
          { std::lock_guard<std::mutex> lk(m_); if (--pending_ == 0) done_.notify_all(); }
        }
      });
    }
  }
  std::mutex busy_, m_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> th_;
  std::function<void(int)> job_;
  int active_ = 0, pending_ = 0;
  uint64_t epoch_ = 0;
};
template <class F>
inline void parallel_ranges(int n, int chunk, F &&fn) {
  const int nthreads = host_threads();
  const int nchunks = (n + chunk - 1) / chunk;
  const int nt = std::min(nthreads, nchunks);
  if (nt <= 1) { if (n > 0) fn(0, n); return; }
  std::atomic<int> next{0};
  auto worker = [&](int) {
    for (int c = next.fetch_add(1); c < nchunks; c = next.fetch_add(1)) fn(c * chunk, std::min(n, (c + 1) * chunk));
  };
  if (!HostPool::get().run(nt - 1, worker)) worker(-1);
}

// A rider item: ops [o0, o0 + n) of target block t, applied by spare workgroups of a k_panel_tri launch of an EARLIER level
// (symbolic.cpp "riders"); first = the target starts from H (+ lambda), otherwise from its partial value in L
struct RideItem { int t, task; long long o0; int n, first; };   // first: 1 = start from H (+ lambda), 0 = from the value in L, 2 = a hub target's piece: t is a scratch block, it receives + sum L_a L_b^T

// undirected block graph of the free poses (CSR, no self loops, no duplicates)
struct BlockGraph {
  int n = 0;
  std::vector<int> xadj, adj;
};

// Output of the symbolic phase: everything the device kernels need that depends only on structure.
constexpr int PANEL_MAX = 16;    // columns per panel; measured on cfg 2: 12 -> 73.1, 16 -> 72.9, 24 -> 67.3, 32 -> 53.5 it/s
// (Round 5 built a second width -- 32 columns for the narrow top levels -- through the symbolic phase, the tables and the triangle / row /
//  forward / backward kernels: 19 instead of 26 levels, but 15 % slower, the pivot chain is per column, not per level.  Removed in round 6;
//  profiles/NOTES.md "32-column panels".)
struct Symbolic {
  int cus = 256;                     // INPUT: compute units of the context's device (256 on an MI355X)
  int nb = 0;                        // block columns (= free poses)
  std::vector<int> perm, iperm;      // perm[k] = hessian index eliminated k-th
  std::vector<int> parent;           // block elimination tree
  std::vector<int64_t> colptr;       // nb+1; column k = [diag, ascending off-diagonal rows]
  std::vector<int> rowidx;           // nnzL
  std::vector<int> blkcol;           // nnzL: column of each block
  // left-looking update lists: L[t] = A[t] - sum_{o in ops(t)} L[op_a[o]] * L[op_b[o]]^T
  std::vector<int64_t> op_ptr;       // nnzL+1
  IntList op_a, op_b;
  std::vector<int64_t> op_mid;       // nnzL: ops [op_ptr, op_mid) external, [op_mid, op_ptr+1) internal
  std::vector<int64_t> acc_ptr;      // nlevels+1 into acc_targets
  std::vector<int> acc_targets;      // blocks with external ops, grouped by level
  std::vector<int64_t> acc_mid;      // nlevels: within a level [acc_ptr, acc_mid) short source lists, [acc_mid, acc_ptr+1) long
  // column-group accumulate lists (k_chol_acc2): the targets of one column that have external updates, in groups of
  // ACC2_G; per group the external source columns j that touch it, ascending, as (b = block (k, j); a[g] = block (i_g, j)
  // or the zero block).  b is the same for the whole wave (scalar loads), every update reads ONE 288-byte block instead
  // of two, and the updates of a target still arrive in ascending source order (bit-identical to the gather lists).
  std::vector<int64_t> g2_lvl;       // nlevels+1 -> groups
  std::vector<int> g2_tgt;           // groups * ACC2_G: target block or -1
  std::vector<int64_t> g2_ptr;       // groups+1 -> entries
  IntList g2_b;                      // entries
  IntList g2_a;                      // entries * ACC2_G
  // row structure (forward solve): row k = blocks L_kj, j < k
  std::vector<int64_t> rowptr;       // nb+1
  std::vector<int> row_blk, row_col;
  // schedule: tasks (sequences of columns run by one workgroup), grouped in dependency levels
  std::vector<int> task_ptr, task_cols;   // task t = task_cols[task_ptr[t] .. task_ptr[t+1])
  std::vector<int> level_ptr;             // level l = tasks [level_ptr[l] .. level_ptr[l+1])
  // ---- panels: a task whose columns c_0 < ... < c_{m-1} (m <= PANEL_MAX) form a path of the elimination tree is a
  // supernode-like PANEL: dense lower triangle {(c_r, c_k)} plus off-triangle rows = the off-diagonal rows of the
  // last column (column patterns are nested along the path).  Levels made of panels only run the panel kernels.
  std::vector<int> task_panel;            // ntask: panel id or -1
  int n_panels = 0;
  std::vector<int> panel_task;            // n_panels
  static size_t tri_off(int pn) { return (size_t)pn * PANEL_MAX * PANEL_MAX; }
  static size_t row_off(int q) { return (size_t)q * PANEL_MAX; }
  static size_t col_off(int pn) { return (size_t)pn * PANEL_MAX; }
  std::vector<int> ptri_blk;              // tri_off(pn) + r * PANEL_MAX + k: block id of (c_r, c_k), r >= k, or -1
  std::vector<int> prow_ptr;              // n_panels+1 -> rows
  std::vector<int> prow_idx;              // per row: its block row index i
  std::vector<int> prow_blk;              // per row * PM: block id of (i, c_k) or -1
  std::vector<char> level_panel;          // nlevels
  std::vector<char> level_leaf;           // nlevels: every task a self-contained light sub-tree -> k_chol_leaf (LDS-resident)
  std::vector<int> level_leaf_maxblk;     // nlevels: most blocks of L in one task of a leaf level
  std::vector<int> level_leaf_maxops;     // nlevels: most update ops in one task of a leaf level
  std::vector<int> pchunk_ptr;            // nlevels+1 -> chunks of <= PANEL_ROWS rows of one panel
  std::vector<int> pchunk_panel, pchunk_row0, pchunk_nrows;
  std::vector<int> panel_chunk0;          // n_panels+1: chunk range of a panel
  std::vector<int> rchunk_ptr;            // nlevels+1 -> chunks of 16 SCALAR rows (one MFMA tile) of one panel
  std::vector<int> rchunk_panel, rchunk_s0;
  // forward solve of panel columns: row list split [external | in-panel]; external part cut into chunks
  std::vector<int64_t> row_mid;           // nb (= rowptr[k+1] for columns outside panels)
  std::vector<int> fchunk_ptr;            // nlevels+1
  std::vector<int> fchunk_col;            // per chunk: column
  std::vector<int64_t> fchunk_e0;         // per chunk: first entry (FWD_CHUNK entries, clipped at row_mid)
  std::vector<int> pcol_fchunk0, pcol_fchunkn;   // n_panels*PM: chunk range of the panel's k-th column
  // ---- riders: early parts of the accumulate of the narrow top levels, run by spare workgroups of earlier triangle launches
  std::vector<RideItem> ride_items;       // grouped by the level whose k_panel_tri launch carries them
  std::vector<int> ride_ptr;              // 2 nlevels + 1 -> ride_items: level l's triangle launch carries [2l, 2l+1), its row launch [2l+1, 2l+2)
  int n_scratch = 0;                      // scratch blocks behind L (index nnzL + 2 + k; nnzL + 1 is an identity block): partial sums of hub targets' pieces
  std::vector<int64_t> acc_start;         // parallel to acc_targets (empty: no riders): first op left to the level's own accumulate launch
                                          // (the target's value so far sits in L), or -1 = the whole list, from H
  // multi-GPU domain decomposition (world > 1): columns are ordered [domain of rank 0 | ... | rank world-1 | top];
  // group g owns columns [dom_col0[g], dom_col0[g+1]), the top is group `world`.  A schedule level is a segment
  // (dependency level, group): seg_group[l].
  int world = 1;
  std::vector<int> dom_col0;              // world + 2
  std::vector<int> seg_group;             // nlevels
  // stats
  int64_t nnzL = 0, nops = 0;
  int etree_height = 0;
  int max_col_blocks = 0;
};


constexpr int ACC2_G = 10;       // targets per group of the column-group accumulate (one wave = 10 lane groups of 6)
constexpr int PANEL_ROWS = 10;   // off-triangle rows per workgroup of the panel row kernels (one wave = 10 lane groups)
// panel levels with more panels than this run the 8-wave k_panel_tri (two workgroups per CU); the others the 16-wave one,
// whose launches can carry rider workgroups
inline int tri_wide_panels(int cus) { return (int)tune("tri_wide", cus); }
constexpr int ACC_LONG_OPS = 64;  // an accumulate target with more external ops than this gets a whole workgroup
constexpr int LEAF_BLOCKS = 168;  // blocks of L a light sub-tree may have (LDS of the leaf kernel): swept on the round-4 schedule, 120 / 144 / 156 / 168 / 176 / 216 -> cfg 2 162.5 / 161.0 / 163.0 / 163.4 / 161.5 / ~159 it/s, cfg 4 factor sweep 2.97 (144) / 2.93 (168) / 3.09 (176) ms, cfg 5 38.7 (144) / 39.0 (168)
constexpr int LEAF_OPS = 3500;    // update ops a light sub-tree may have (4 B each in LDS next to its blocks: 2 workgroups per CU stay possible)
constexpr int ROW_SETS = 1;       // k_panel_rows: sets of 16 scalar rows per wave.  2 was measured: every operand tile load feeds two MFMAs, but 196 VGPRs leave one wave per SIMD and the latency-bound top levels lose more (factor sweep 5.94 -> 6.53 ms)
constexpr int FWD_CHUNK = 320;   // row-list entries per workgroup of the wide forward-solve kernel

struct OrderingOptions {
  int leaf = 64;
  // cut score = separator size x (1 + bal_w max(0, imbalance - bal_t)).  bal_w: 8 until round 4; 5 measured better on every
  // 100k-pose seed tried (42 / 43 / 44 / 7: +2.2 / +8.1 / +4.7 / +4.5 % iterations/s, 1-4 levels fewer) and neutral on the 1M-pose
  // graph, the BA / VIO graphs, the torus and the hub graph (profiles/NOTES.md round 4)
  double bal_w = 5.0, bal_t = 0.35;
  // vertices with degree > max(dense_min, dense_factor * mean degree) are hubs: eliminated last (0 disables)
  double dense_factor = 10.0;
  int dense_min = 64;
  // time dissection (graphs whose vertex index is time; ordering.cpp split_by_index): smaller side of a cut >= time_side of the
  // region, measured in vertices (time_weight = 0) or in 1 + time_weight x crossing edges (loop-dense stretches count for more)
  double time_side = 0.30, time_weight = 0.0;
  // time dissection under RECOVERED labels (round 6: one Cuthill-McKee pass when the vertex index is not time).  Off for graphs of mixed
  // variable kinds: on the VIO graph (X / V / B keyed by type) the recovered order passes the band test but its index cuts give 30 levels
  // against the 27 of the level structures (host, tools/symstats on cfg 4's edges) -- pose graphs only.
  bool time_recover = true;
};

void nested_dissection(const BlockGraph &g, const OrderingOptions &opt, std::vector<int> &perm);
void build_symbolic(const BlockGraph &g, const std::vector<int> &perm, int64_t task_work_limit, int64_t chain_work_limit, Symbolic &S, int world = 1,
                    bool analyse_only = false);   // analyse_only: stop once fill, block updates and levels are known (a fifth of the time)

}  // namespace fgo
