// The RCCL types pysdr_amd/csrc/api.hip names (it dlopens the library; tests/host_san never does)
#pragma once
#include <hip/hip_runtime.h>
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint8 = 1 } ncclDataType_t;
