// shim: the GTSAM slice used by the graph wrappers lives in ../../gtsam_lite.h (GTSAM is not installed in this image)
#include "../../gtsam_lite.h"
