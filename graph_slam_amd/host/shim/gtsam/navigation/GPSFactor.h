// include-path forwarder: the GTSAM API slice lives in shim/gtsam_lite.h
#pragma once
#include "../../gtsam_lite.h"
