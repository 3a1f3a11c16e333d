// TEST INFRASTRUCTURE: the few RCCL names mpc_comm.hpp mentions (the emulated library never opens librccl: single rank).
#pragma once
#include <cstddef>
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclChar = 0, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclMax = 2 } ncclRedOp_t;
