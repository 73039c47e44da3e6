// include-path forwarder: the g2o API slice lives in shim/g2o/fgo_g2o.h
#pragma once
#include "../fgo_g2o.h"
