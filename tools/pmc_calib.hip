// Calibration of the rocprofv3 HBM counters (FETCH_SIZE / WRITE_SIZE) on the access patterns libfgo's kernels use
// (the guide's x2 correction is established for wide coalesced streaming reads only; "other access
// widths and WRITE_SIZE are uncalibrated" -- /opt/skills/guides/MI355X_MICROARCH.md "HBM").  Every kernel below moves a KNOWN
// number of bytes exactly once over a 1.2 GB working set (>> 256 MB Infinity Cache, >> 32 MB of L2):
//   cal_read_stream16     16 B per lane, fully coalesced                       (k_linearize's record halves, L copies)
//   cal_read_rows_seq     48-byte rows of consecutive 288-byte blocks, lane = 6 g + r   (apply_ops / load_row, blocks in order)
//   cal_read_rows_rand    the same rows of blocks in a random permutation      (the gather-form accumulate)
//   cal_read_elem8_rand   8-byte elements, a wave covers 16 rows x 24 doubles of random blocks  (k_panel_rows' U gather)
//   cal_write_stream16 / cal_write_rows_rand / cal_write_elem8_rand            the store-side twins
// tools/pmc_calib.sh runs it under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, --kernel-trace only) and
// prints counter / known bytes per pattern; tools/pmc_traffic.py applies the per-pattern factors.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void cal_read_stream16(const double2 *__restrict__ p, long n2, double *out) {
  double acc = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) { const double2 v = p[i]; acc += v.x + v.y; }
  if (acc == 12345.678) out[0] = acc;
}
__global__ void cal_write_stream16(double2 *__restrict__ p, long n2) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) p[i] = make_double2(1.0, 2.0);
}
// one wave = 10 blocks, lane 6 g + r reads / writes row r (3 x 16 B) of block idx[10 w + g]
template <int TAG> __global__ void cal_read_rows(const double *__restrict__ L, const int *__restrict__ idx, long nblk, double *out) {
  const int lane = threadIdx.x & 63, g = lane / 6, r = lane - 6 * g;
  const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  double acc = 0;
  if (lane < 60 && 10 * w + g < nblk) {
    const double *b = L + 36 * (long)idx[10 * w + g] + 6 * r;
    const double2 a0 = *reinterpret_cast<const double2 *>(b), a1 = *reinterpret_cast<const double2 *>(b + 2), a2 = *reinterpret_cast<const double2 *>(b + 4);
    acc = a0.x + a0.y + a1.x + a1.y + a2.x + a2.y;
  }
  if (acc == 12345.678) out[0] = acc;
}
__global__ void cal_write_rows(double *__restrict__ L, const int *__restrict__ idx, long nblk) {
  const int lane = threadIdx.x & 63, g = lane / 6, r = lane - 6 * g;
  const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  if (lane < 60 && 10 * w + g < nblk) {
    double *b = L + 36 * (long)idx[10 * w + g] + 6 * r;
    *reinterpret_cast<double2 *>(b) = make_double2(1, 2); *reinterpret_cast<double2 *>(b + 2) = make_double2(3, 4); *reinterpret_cast<double2 *>(b + 4) = make_double2(5, 6);
  }
}
// k_panel_rows' pattern: a wave covers 16 scalar rows x 16 block columns; lane (nn = lane & 15, q = lane >> 4) reads the 24
// doubles  column c = 16 J + q + 4 r4  of scalar row nn: block idx[16 * (row block) + c / 6], 8 bytes each.  Per wave: 16 rows
// x 96 columns = 1536 doubles = every element of 16/6 x 16 blocks' rows -- here: wave w owns 48 blocks (3 row blocks x 16 columns),
// read completely in two sweeps of 16 scalar rows (rows 0..15 and 2..17 would overlap: use 18 rows = 3 blocks, sweeps of 9)
__global__ void cal_read_elem8(const double *__restrict__ L, const int *__restrict__ idx, long nblk, double *out) {
  const int lane = threadIdx.x & 63, nn = lane & 15, q = lane >> 4;
  const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  double acc = 0;
  if (48 * w + 47 < nblk) {
    for (int half = 0; half < 2; ++half) {
      const int s = 9 * half + nn;                              // scalar row 0 .. 17 of the wave's 3 row blocks
      if (nn < 9) {
        const int br = s / 6, rho = s - 6 * br;
        for (int e = 0; e < 24; ++e) {
          const int c = 16 * (e >> 2) + q + 4 * (e & 3);
          acc += L[36 * (long)idx[48 * w + 16 * br + c / 6] + 6 * rho + (c - 6 * (c / 6))];
        }
      }
    }
  }
  if (acc == 12345.678) out[0] = acc;
}
__global__ void cal_write_elem8(double *__restrict__ L, const int *__restrict__ idx, long nblk) {
  const int lane = threadIdx.x & 63, nn = lane & 15, q = lane >> 4;
  const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  if (48 * w + 47 < nblk) {
    for (int half = 0; half < 2; ++half) {
      const int s = 9 * half + nn;
      if (nn < 9) {
        const int br = s / 6, rho = s - 6 * br;
        for (int e = 0; e < 24; ++e) {
          const int c = 16 * (e >> 2) + q + 4 * (e & 3);
          L[36 * (long)idx[48 * w + 16 * br + c / 6] + 6 * rho + (c - 6 * (c / 6))] = 1.0;
        }
      }
    }
  }
}

int main(int argc, char **argv) {
  const long nblk = argc > 1 ? std::atol(argv[1]) : 4L * 1024 * 1024 + 32;       // x 288 B = 1.2 GB
  const long ndbl = 36 * nblk;
  double *L = nullptr, *out = nullptr; int *idx_seq = nullptr, *idx_rand = nullptr;
  CHK(hipMalloc(&L, sizeof(double) * ndbl)); CHK(hipMalloc(&out, 64));
  CHK(hipMalloc(&idx_seq, sizeof(int) * nblk)); CHK(hipMalloc(&idx_rand, sizeof(int) * nblk));
  std::vector<int> h(nblk); std::iota(h.begin(), h.end(), 0);
  CHK(hipMemcpy(idx_seq, h.data(), sizeof(int) * nblk, hipMemcpyHostToDevice));
  std::mt19937_64 rng(7); std::shuffle(h.begin(), h.end(), rng);
  CHK(hipMemcpy(idx_rand, h.data(), sizeof(int) * nblk, hipMemcpyHostToDevice));
  CHK(hipMemset(L, 0, sizeof(double) * ndbl));
  CHK(hipDeviceSynchronize());
  const long nw10 = (nblk + 9) / 10, nw48 = nblk / 48;
  const long used48 = nw48 * 48;
  std::printf("# pattern kernel known_bytes (index arrays: +4 B per block where used, listed separately)\n");
  hipLaunchKernelGGL(cal_read_stream16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const double2 *>(L), ndbl / 2, out);
  std::printf("read  cal_read_stream16   %ld  idx 0\n", ndbl * 8);
  hipLaunchKernelGGL(cal_read_rows<0>, dim3((nw10 + 3) / 4), dim3(256), 0, 0, L, idx_seq, nblk, out);
  std::printf("read  cal_read_rows<0>    %ld  idx %ld\n", nblk * 288, nblk * 4);
  hipLaunchKernelGGL(cal_read_rows<1>, dim3((nw10 + 3) / 4), dim3(256), 0, 0, L, idx_rand, nblk, out);
  std::printf("read  cal_read_rows<1>    %ld  idx %ld\n", nblk * 288, nblk * 4);
  hipLaunchKernelGGL(cal_read_elem8, dim3((nw48 + 3) / 4), dim3(256), 0, 0, L, idx_rand, nblk, out);
  std::printf("read  cal_read_elem8      %ld  idx %ld\n", used48 * 288, used48 * 4);
  CHK(hipDeviceSynchronize());
  hipLaunchKernelGGL(cal_write_stream16, dim3(4096), dim3(256), 0, 0, reinterpret_cast<double2 *>(L), ndbl / 2);
  std::printf("write cal_write_stream16  %ld  idx 0\n", ndbl * 8);
  hipLaunchKernelGGL(cal_write_rows, dim3((nw10 + 3) / 4), dim3(256), 0, 0, L, idx_rand, nblk);
  std::printf("write cal_write_rows      %ld  idx %ld\n", nblk * 288, nblk * 4);
  hipLaunchKernelGGL(cal_write_elem8, dim3((nw48 + 3) / 4), dim3(256), 0, 0, L, idx_rand, nblk);
  std::printf("write cal_write_elem8      %ld  idx %ld\n", used48 * 288, used48 * 4);
  CHK(hipDeviceSynchronize());
  CHK(hipGetLastError());
  return 0;
}
