// Internal host-side data structures of libfgo (not part of the C-ABI).
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace fgo {

// undirected block graph of the free poses (CSR, no self loops, no duplicates)
struct BlockGraph {
  int n = 0;
  std::vector<int> xadj, adj;
};

// Output of the symbolic phase: everything the device kernels need that depends only on structure.
struct Symbolic {
  int nb = 0;                        // block columns (= free poses)
  std::vector<int> perm, iperm;      // perm[k] = hessian index eliminated k-th
  std::vector<int> parent;           // block elimination tree
  std::vector<int64_t> colptr;       // nb+1; column k = [diag, ascending off-diagonal rows]
  std::vector<int> rowidx;           // nnzL
  std::vector<int> blkcol;           // nnzL: column of each block
  // left-looking update lists: L[t] = A[t] - sum_{o in ops(t)} L[op_a[o]] * L[op_b[o]]^T
  std::vector<int64_t> op_ptr;       // nnzL+1
  std::vector<int> op_a, op_b;
  std::vector<int64_t> op_mid;       // nnzL: ops [op_ptr, op_mid) external, [op_mid, op_ptr+1) internal
  std::vector<int64_t> acc_ptr;      // nlevels+1 into acc_targets
  std::vector<int> acc_targets;      // blocks with external ops, grouped by level
  // row structure (forward solve): row k = blocks L_kj, j < k
  std::vector<int64_t> rowptr;       // nb+1
  std::vector<int> row_blk, row_col;
  // schedule: tasks (sequences of columns run by one workgroup), grouped in dependency levels
  std::vector<int> task_ptr, task_cols;   // task t = task_cols[task_ptr[t] .. task_ptr[t+1])
  std::vector<int> level_ptr;             // level l = tasks [level_ptr[l] .. level_ptr[l+1])
  // stats
  int64_t nnzL = 0, nops = 0;
  int etree_height = 0;
  int max_col_blocks = 0;
};

struct OrderingOptions {
  int leaf = 64;
  // vertices with degree > max(dense_min, dense_factor * mean degree) are hubs: eliminated last (0 disables)
  double dense_factor = 10.0;
  int dense_min = 64;
};

void nested_dissection(const BlockGraph &g, const OrderingOptions &opt, std::vector<int> &perm);
void build_symbolic(const BlockGraph &g, const std::vector<int> &perm, int64_t task_work_limit, int64_t chain_work_limit, Symbolic &S);

}  // namespace fgo
