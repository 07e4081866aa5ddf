// oracle/ref_shim/IMU/imudata.h -- TEST INFRASTRUCTURE: stands in for src/IMU/imudata.h so that include/Frame.h compiles (the IMU code paths of
// src/Frame.cc are compiled, never executed).
#ifndef YGZ_ORACLE_REF_SHIM_IMUDATA_H
#define YGZ_ORACLE_REF_SHIM_IMUDATA_H
#include "frame_stubs.h"
#endif
