// include-path forwarder (shim/qt_lite.h)
#pragma once
#include "qt_lite.h"
