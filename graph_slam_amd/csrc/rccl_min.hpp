// The few RCCL types libfgo needs, declared locally: librccl is resolved with dlopen at run time (fgo_dist.cpp), so
// libfgo builds on a ROCm install without the RCCL development headers.  Values as in rccl/rccl.h (NCCL ABI).
#pragma once
#include <stddef.h>

extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;          // anything else is a failure; the text comes from ncclGetErrorString
typedef enum { ncclDouble = 8 } ncclDataType_t;         // ncclFloat64
typedef enum { ncclSum = 0, ncclMax = 2 } ncclRedOp_t;
}
