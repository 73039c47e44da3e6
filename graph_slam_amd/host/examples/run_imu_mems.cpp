// Harness in the builder's own words for the reference's MEMS IMU interface, compiled IN PLACE from
// /root/reference/gtsam/imu_MEMS.cpp (CImuMEMS: the `id1 gx gy gz ax ay az id2` integer log, counts -> rad/s and m/s^2,
// prior bias from the stationary samples before the camera synchronisation point, gravity 9.81, dt = 10 ms,
// gtsam/imu_MEMS.cpp:9-13,22-97,99-163).  VERDICT r2 missing #7: it compiled but was never exercised.
//   usage: run_imu_mems <mems.log> <next_i> [<i> <j>]
// prints one JSON line: synchronisation index, prior bias, the preintegrated payload after predictNext(next_i), the
// predicted state, and (optionally) predictBetween(i, j) from that state.  Host-only: needs no GPU.
#include <cstdio>
#include <cstdlib>
#include <string>
#include "imu_MEMS.h"

using namespace gtsam;

static void print_pose(const char *name, const NavState &s) {
  const Quaternion q = s.pose().rotation().toQuaternion();
  std::printf("\"%s\": [%.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g, %.17g]", name, s.pose().x(), s.pose().y(), s.pose().z(), q.x(), q.y(), q.z(), q.w(),
              s.v()(0), s.v()(1), s.v()(2));
}

int main(int argc, char **argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s <mems.log> <next_i> [<i> <j>]\n", argv[0]); return 1; }
  CImuMEMS imu;
  if (!imu.readImuData(argv[1])) return 2;
  imu.computePriorBias();
  const NavState s1 = imu.predictNext(std::atoi(argv[2]));
  const PreintegratedCombinedMeasurements *pim = dynamic_cast<PreintegratedCombinedMeasurements *>(imu.mp_combined_pre_imu);
  if (!pim) return 3;
  const fgo_preint &p = pim->raw();
  std::printf("{\"syn_start\": %d, \"n\": %zu, \"dt\": %.17g, \"bias\": [", imu.m_syn_start_id, imu.mv_measurements.size(), (double)imu.m_dt);
  const Vector6 b = imu.m_prior_imu_bias.vector();
  for (int k = 0; k < 6; ++k) std::printf("%s%.17g", k ? ", " : "", b(k));
  std::printf("], \"payload\": [");
  const double *pd = reinterpret_cast<const double *>(&p);
  for (size_t k = 0; k < sizeof(fgo_preint) / sizeof(double); ++k) std::printf("%s%.17g", k ? ", " : "", pd[k]);
  std::printf("], ");
  print_pose("predict_next", s1);
  if (argc >= 5) {
    NavState st = s1;
    const NavState s2 = imu.predictBetween(std::atoi(argv[3]), std::atoi(argv[4]), st);
    std::printf(", ");
    print_pose("predict_between", s2);
  }
  std::printf(", \"first\": [");
  for (int k = 0; k < 6; ++k) std::printf("%s%.17g", k ? ", " : "", imu.mv_measurements[0](k));
  std::printf("]}\n");
  return 0;
}
